// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
//
// CPU restatement of the reference's decentralized-optimization phase (sqp/ subsystem):
//   sqp/inter_agent_cons.cc  (bridge: interpolation, neighbour pairs, separating planes)
//   sqp/corridor.cc          (axis-aligned safe boxes)
//   sqp/dsqp_solver.cc       (per-agent SQP: linearise, assemble QP, OSQP, feasibility, corridor refresh)
//   sqp/utils.cc             (extractResult, readQpSolverConfig arithmetic)
//   common/motion_planning.h (State float disc centres, agentDistance, agentCollision)
// Each function cites the reference lines it follows.  PARITY UNPINNED — see osqp_restate.h.
#pragma once
#include <array>
#include <cstdint>
#include <vector>

#include "osqp_restate.h"

namespace csdo_oracle {

// Constants::* of the reference are `float` statics (common/motion_planning.h:12-49); keep them float here so that
// every promotion to double happens where the reference's does.
struct Vehicle {
  float r = 3.f, deltat = 0.706f;
  float LF = 2.f, LB = 1.f, carWidth = 2.f, WB = 1.f;
  float f2x = 1.25f, r2x = -0.25f, rv = 1.25f;
  float obsRadius = 0.8f;
  void derive();  // common/motion_planning.cc:82-85
};

struct QpParm {  // sqp/common.h:39-52
  double r_trust = 2.0, max_omega = 0.07, max_v = 1.0, max_iter = 10, delta_solution_threshold = 1.0,
         max_violation = 1e-3;
  int osqp_max_iter = 400;
  double dt = 0.0;
  int num_interpolation = 2;
  bool fixed_corridor = false;
  int adaptive_rho_interval = 25;  // restatement parameter (osqp_restate.h header note)
};
double qp_dt(const Vehicle& v, double max_v, int num_interpolation, double decelerate_factor);  // utils.cc:55-56

struct OptRes {  // sqp/common.h:14-22
  double x = 0, y = 0, yaw = 0, v = 0, a = 0, steer = 0, d_steer = 0;
};
struct InterPlane {  // sqp/inter_agent_cons.h:47-63; c[] = a_f2f,b_f2f,c_f2f, a_f2r,.., a_r2f,.., a_r2r,b_r2r,c_r2r
  int t = 0;
  double c[12] = {0};
};
struct Obstacle {
  double x, y, r;
};
struct Corridor {  // sqp/corridor.h:8-11
  double xf_min, xf_max, yf_min, yf_max, xr_min, xr_max, yr_min, yr_max;
};
struct Box {
  double x_min, y_min, x_max, y_max;
};
struct BoxStatus {
  bool success;
  int initial_status;
};

// ---- common/motion_planning.h State restated ----
struct DiscCentres {
  float xf, yf, xr, yr, xc, yc;
};
DiscCentres state_discs(double x, double y, double yaw, const Vehicle& v);                 // :115-132
double agent_distance(const DiscCentres& a, const DiscCentres& b);                         // :208-217
bool agent_collision(const DiscCentres& a, double yaw_a, const DiscCentres& b, double yaw_b,
                     const Vehicle& v);                                                      // :140-183
float normalize_angle_abs_in_pi(double x);                                                  // :70-75

// ---- sqp/inter_agent_cons.cc restated ----
struct CoarsePath {               // PlanResult<State,Action,double> reduced to what the bridge reads
  std::vector<std::array<double, 3>> states;  // x, y, yaw
  std::vector<int> actions;                   // 0..6, size states-1
};
void interpolate_initial_guess(std::vector<CoarsePath> paths, const std::vector<std::array<double, 3>>& goals,
                               const QpParm& parm, const Vehicle& veh,
                               std::vector<std::vector<OptRes>>& x0_bar);                    // :143-157
bool find_neighbor_pairs(const std::vector<std::vector<OptRes>>& sol, double r_trust, const Vehicle& veh,
                         std::vector<std::array<int, 3>>& pairs);                           // :12-49
void calc_inter_planes(const std::vector<std::vector<OptRes>>& x0_bar,
                       const std::vector<std::array<int, 3>>& pairs, const Vehicle& veh,
                       std::vector<std::vector<InterPlane>>& planes);                       // :71-140

// ---- sqp/corridor.cc restated ----
bool is_box_valid(const Box& b, const std::vector<Obstacle>& obs, double dimx, double dimy, const Vehicle& v);
bool generate_local_box(double xc, double yc, const std::vector<Obstacle>& obs, double dimx, double dimy,
                        const Vehicle& v, Box& res, double ds = 0.1, double l_limit = 10.0);
BoxStatus generate_box(double dimx, double dimy, double x, double y, const std::vector<Obstacle>& obs,
                       const Vehicle& v, Box& res);
bool calc_corridors(const std::vector<std::vector<OptRes>>& guesses, const std::vector<Obstacle>& obs,
                    double dimx, double dimy, const Vehicle& v, std::vector<std::vector<Corridor>>& corridors,
                    double& time_max_corridor);

// ---- sqp/dsqp_solver.cc restated ----
struct AgentQp {  // one SQP iteration's QP in the reference's field-major layout
  Csc P_triu, A;
  std::vector<double> q, l, u;
};
// Assemble the QP of agent-local data at linearisation point sol0 (length 6Nt-2).  corr_lb/ub: [4][Nt]
// (xf,yf,xr,yr); x_trust/y_trust: original initial guess; cfg[6].
void assemble_qp(int Nt, const std::vector<double>& sol0, const std::vector<double>& corr_lb,
                 const std::vector<double>& corr_ub, const std::vector<double>& x_trust,
                 const std::vector<double>& y_trust, const double cfg[6],
                 const std::vector<InterPlane>& planes, const QpParm& parm, const Vehicle& veh, AgentQp& qp);

struct DsqpResult {
  std::vector<std::vector<OptRes>> solutions;      // [Na][Nt]
  std::vector<std::vector<Corridor>> corridors;    // final boxes
  std::vector<int> sqp_iters, admm_iters, last_status;
  std::vector<double> agent_seconds;
  int solver_status = 1;
  bool initial_static_legal = true;
  double t_total = 0, t_max_individual = 0, t_corridor_max = 0;
};

// Optional per-agent, per-SQP-iteration record for golden fixtures.
struct SqpTraceEntry {
  int agent, sqp_iter, status, admm_iter;
  double delta;
  std::vector<double> sol;  // 6Nt-2
};

struct DsqpProblem {
  std::vector<std::vector<OptRes>> x0_bar;
  std::vector<std::vector<InterPlane>> planes;
  double dimx = 0, dimy = 0;
  std::vector<Obstacle> obstacles;
  QpParm parm;
  Vehicle veh;
};

// SolverDSQP::SolverDSQP restated (dsqp_solver.cc:1133-1249).  n_threads = 1 reproduces the serial agent loop of
// :1198-1220; n_threads > 1 solves one agent per thread (the "fair parallel CPU" baseline of SURVEY 8d).
void dsqp_solve(const DsqpProblem& prob, DsqpResult& res, int n_threads = 1,
                std::vector<SqpTraceEntry>* trace = nullptr);

// The same for several independent worlds with one thread pool over all their agents (bench.py's cpu_baseline).
void dsqp_solve_batch(const std::vector<DsqpProblem>& probs, std::vector<DsqpResult>& results, int n_threads);

}  // namespace csdo_oracle
