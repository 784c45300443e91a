// Plain-old-data descriptors shared by the host launcher and the device lane program.
// Layout in HBM (all fp64 unless noted), one launch = a batch of agents, one workgroup per agent:
//   x0        [sum_a Nt_a][6]     initial guess x,y,yaw,steer,v,d_steer          (read once per agent)
//   planes    [sum_a K_a]         {t, c[12]} per agent, t ascending               (read once per SQP iteration)
//   tstart    [sum_a (Nt_a+1)]    CSR offsets of an agent's planes by timestep    (int32)
//   obstacles [sum_w n_obs_w][3]  per world                                       (staged into LDS)
//   rows_ws   [sum_a][8][4K_a]    inter-vehicle row state by field (y,z,dy | u,E,ca,cb,cyaw), L2-resident
//   fac_ws    [sum_a 357*Nt_a]    BCR coupling blocks (when not in LDS) + factor exchange + per-lane workspace
//   sol       [sum_a Nt_a][6], corr [sum_a Nt_a][8], per-agent counters           (written once per agent)
#pragma once
#include <cstdint>

namespace csdo {

struct AgentDesc {
  int32_t Nt;          // horizon of this agent's world
  int32_t world;       // index into WorldDesc
  int32_t n_planes;    // K_a
  int32_t rows_lds;    // 1: the inter-vehicle rows' duals / slacks / rhs shares fit LDS beside the rest (set by the launcher)
  int32_t tail_nodes;  // capacity of the dense tail of the BCR solve in 6x6 nodes: 6, or 8 / 12 (512-thread class; dsqp_class.h: a
                       // function of the agent alone, the lane-serial build applies the same rule); 0 = 6
  int32_t pad_;
  int64_t x0_off;      // element offset of x0[Nt][6] (in doubles)
  int64_t plane_off;   // first plane of this agent
  int64_t tstart_off;  // offset of tstart[Nt+1]
  int64_t rows_off;    // first inter row (4 per plane) in rows_ws, in rows
  int64_t fac_off;     // offset into fac_ws (doubles)
  int64_t out_off;     // timestep offset into sol / corr
};

struct WorldDesc {
  double dimx, dimy;
  int32_t obs_off, n_obs;
};

struct PlaneDev {  // mirrors csdo_plane
  int32_t t, _pad;
  double c[12];
};

struct SolverParams {
  // vehicle (float-rounded values widened to double; common/motion_planning.h:12-49 of the reference)
  double f2x, r2x, rv, WB, r_turn;
  double steer_max;   // atan(WB / r_turn) (sqp/dsqp_solver.cc:1178), a constant of the batch: formed once by the host
  // QpParm (sqp/common.h:39-52)
  double r_trust, max_omega, max_v, delta_solution_threshold, dt;
  int32_t max_iter, osqp_max_iter, fixed_corridor, adaptive_rho_interval;
  int32_t solve_refinement, pad_;   // csdo_qp_parm::solve_refinement: which kernel instantiations a launch uses (REFINE)
  // OSQP 0.6.3 defaults (osqp_set_default_settings)
  double rho0, sigma, alpha, eps_abs, eps_rel, eps_prim_inf;
  int32_t scaling_passes, check_termination;
  double adaptive_rho_tolerance;
};

struct DeviceBatch {
  const AgentDesc* agents;
  const WorldDesc* worlds;
  const double* x0;
  const PlaneDev* planes;
  const int32_t* tstart;
  const double* obstacles;
  double* rows_ws;
  double* fac_ws;
  double* sol;
  double* corr;
  int32_t* sqp_iters;
  int32_t* admm_iters;
  int32_t* last_status;
  int32_t* static_legal;   // per agent: 1 if every initial box was legal
  int64_t* agent_ticks;    // per agent device time, 100 MHz ticks
  const int32_t* order;    // launch order: agents grouped by kernel class, heaviest first inside a group
  int32_t n_agents;
  int32_t _pad;
  int64_t* prof;           // diagnostic builds only: [n_agents][16] shader-clock ticks per phase, else null
  SolverParams prm;
};

constexpr int ROWS_WS_STRIDE = 9;   // doubles per inter row in rows_ws: 8 fields + room for the per-plane rhs shares
constexpr int FAC_E_DOUBLES = 108;  // per lane, lane-major: [0,36) and [72,108) the two 6x6 blocks the LANE multiplies in the solve (pair-split
                                    // modes: dsqp_program.h; the long-horizon modes keep F_l of the NODE in [0,36)), [36,72) the node's coupling R / F_r
constexpr int FAC_X_DOUBLES = 100;  // factor-time exchange per node: U_l(21) + U_r(21) + Rnew(36) + diagonal block (21, padded)
constexpr int COLD_DOUBLES = 210;   // per-lane workspace slots (WsSlot in dsqp_program.h)

}  // namespace csdo
