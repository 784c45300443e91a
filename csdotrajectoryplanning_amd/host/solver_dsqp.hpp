// solver_dsqp.hpp — header-only C++ mirror of the reference's `SolverDSQP` (sqp/dsqp_solver.h:24-47) on top of the C ABI
// (include/csdo_dsqp.h).  Same constructor shape, same public members and getters, so csdo.cc keeps its call site
// (csdo.cc:146-159).  The constructor is a template over the caller's element types: it reads the FIELDS of whatever
// OptimizeResult / InterPlane / Location / QpParm types it is handed (the reference's own
// libMultiRobotPlanning structs, sqp/common.h:14-52, sqp/inter_agent_cons.h:47-63, common/motion_planning.h:79-84, pass
// unchanged - no conversion, no typedef), and any iterable of obstacles (the reference passes an unordered_set).
// tests/cpp/mirror_main.cc compiles this header against stand-ins shaped like the reference's structs.
// Link with -lcsdo_hip.
#pragma once
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/csdo_dsqp.h"

namespace csdo {

struct OptimizeResult {
  double x = 0, y = 0, yaw = 0, v = 0, a = 0, steer = 0, d_steer = 0;
};
struct InterPlane {
  int t;
  double a_f2f, b_f2f, c_f2f, a_f2r, b_f2r, c_f2r, a_r2f, b_r2f, c_r2f, a_r2r, b_r2r, c_r2r;
};
struct Corridor {
  double xf_min, xf_max, yf_min, yf_max, xr_min, xr_max, yr_min, yr_max;
};
struct Location {
  double x, y, r;
};
struct QpParm {
  double r_trust, max_omega, max_v, max_iter, delta_solution_threshold, max_violation;
  int osqp_max_iter;
  double dt;
  int num_interpolation;
  bool fixed_corridor;
};

class SolverDSQP {
 public:
  // OptRes: fields x, y, yaw, v, steer, d_steer (sqp/common.h:14-22).  Plane: t, a_f2f .. c_r2r (inter_agent_cons.h:47-63).
  // Parm: the QpParm fields (sqp/common.h:39-52).  ObstacleRange: any iterable whose elements have x, y, r (the reference
  // passes an unordered_set<Location>; order only matters when a point lies inside two inflated obstacles - this backend
  // uses the iteration order it is given).  The structs above are conveniences for callers that have no types of their own.
  template <class OptRes, class Plane, class ObstacleRange, class Parm>
  SolverDSQP(std::vector<std::vector<OptRes>>& solutions, const std::vector<std::vector<OptRes>>& x0_bar,
             const std::vector<std::vector<Plane>>& inter_planes, double dimx, double dimy,
             const ObstacleRange& obstacles, const Parm& param, int logger_level = 2, int device = 0,
             const csdo_vehicle* vehicle = nullptr, const std::vector<int>& devices = {}, bool solve_refinement = false) {
    // solve_refinement: csdo_qp_parm::solve_refinement - every linear solve refined on the KKT residual (as accurate as OSQP's LDL',
    // about twice the kernel time); no reference counterpart
    // devices: several GPU ordinals - the agents are cut into contiguous blocks of equal estimated work, one per device
    // (csdo_dsqp_create_multi; the loop that shards is sqp/dsqp_solver.cc:1198-1220); empty: the one GPU `device`
    const int Na = (int)x0_bar.size();
    const int Nt = Na ? (int)x0_bar[0].size() : 0;
    std::vector<double> x0((size_t)Na * Nt * 6), obs;
    for (int a = 0; a < Na; ++a)
      for (int t = 0; t < Nt; ++t) {
        const OptRes& r = x0_bar[a][t];
        double* g = &x0[((size_t)a * Nt + t) * 6];
        g[0] = r.x; g[1] = r.y; g[2] = r.yaw; g[3] = r.steer; g[4] = r.v; g[5] = r.d_steer;
      }
    std::vector<int32_t> off(Na + 1, 0);
    std::vector<csdo_plane> planes;
    for (int a = 0; a < Na; ++a) {
      for (const Plane& p : inter_planes[a]) {
        csdo_plane q{};
        q.t = p.t;
        const double c[12] = {p.a_f2f, p.b_f2f, p.c_f2f, p.a_f2r, p.b_f2r, p.c_f2r,
                              p.a_r2f, p.b_r2f, p.c_r2f, p.a_r2r, p.b_r2r, p.c_r2r};
        for (int k = 0; k < 12; ++k) q.c[k] = c[k];
        planes.push_back(q);
      }
      off[a + 1] = (int32_t)planes.size();
    }
    for (const auto& o : obstacles) {
      obs.push_back(o.x);
      obs.push_back(o.y);
      obs.push_back(o.r);
    }
    csdo_problem P{};
    P.Na = Na;
    P.Nt = Nt;
    P.x0_bar = x0.data();
    P.plane_off = off.data();
    P.planes = planes.data();
    P.dimx = dimx;
    P.dimy = dimy;
    P.n_obs = (int32_t)(obs.size() / 3);
    P.obstacles = obs.data();
    if (vehicle) P.veh = *vehicle; else csdo_vehicle_default(&P.veh);
    P.parm.r_trust = param.r_trust;
    P.parm.max_omega = param.max_omega;
    P.parm.max_v = param.max_v;
    P.parm.max_iter = (double)param.max_iter;
    P.parm.delta_solution_threshold = param.delta_solution_threshold;
    P.parm.max_violation = param.max_violation;
    P.parm.osqp_max_iter = param.osqp_max_iter;
    P.parm.num_interpolation = param.num_interpolation;
    P.parm.dt = param.dt;
    P.parm.fixed_corridor = param.fixed_corridor ? 1 : 0;
    P.parm.adaptive_rho_interval = 0;  // documented default (25)
    P.parm.solve_refinement = solve_refinement ? 1 : 0;
    P.parm._reserved = 0;
    P.logger_level = logger_level;

    std::vector<double> sol((size_t)Na * Nt * 6), cor((size_t)Na * Nt * 8);
    std::vector<int32_t> admm(Na), last(Na);
    num_iterations.assign(Na, 0);
    csdo_result R{};
    R.solutions = sol.data();
    R.corridors = cor.data();
    R.sqp_iters = num_iterations.data();
    R.admm_iters = admm.data();
    R.last_status = last.data();
    std::vector<double> agent_s((size_t)Na, 0.0);
    R.agent_seconds = agent_s.data();
    csdo_handle h = nullptr;
    const std::vector<int32_t> devs(devices.begin(), devices.end());
    int rc = devs.empty() ? csdo_dsqp_create(&h, device) : csdo_dsqp_create_multi(&h, devs.data(), (int32_t)devs.size());
    if (rc == CSDO_OK) rc = csdo_dsqp_solve(h, &P, &R);
    if (h) csdo_dsqp_destroy(h);
    if (rc != CSDO_OK) throw std::runtime_error("csdo_dsqp_solve failed with code " + std::to_string(rc));

    solutions.assign(Na, std::vector<OptRes>(Nt));
    corridors.assign(Na, std::vector<Corridor>(Nt));
    for (int a = 0; a < Na; ++a)
      for (int t = 0; t < Nt; ++t) {
        const double* s = &sol[((size_t)a * Nt + t) * 6];
        OptRes& r = solutions[a][t];
        r.x = s[0]; r.y = s[1]; r.yaw = s[2]; r.steer = s[3]; r.v = s[4]; r.d_steer = s[5];
        const double* c = &cor[((size_t)a * Nt + t) * 8];
        corridors[a][t] = Corridor{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]};
      }
    admm_iterations.assign(admm.begin(), admm.end());
    // Diagnostics.  The reference's logger_level >= 2 prints stage timers and the step size per agent and SQP iteration while it
    // solves (sqp/dsqp_solver.cc:116-120,199-202,229-236,504-508), level >= 3 the iteration count (:258-260).  Here every agent's whole
    // SQP runs on the device in one launch and nothing is printed from there; what the host knows afterwards is printed instead, one
    // line per agent (per-stage cycle counts: the phase-timer build, scripts/profile_phases_sum.py).
    if (logger_level >= 2) {
      std::printf("csdo::SolverDSQP: %d agents, Nt = %d, kernels %.3f ms, call %.3f ms, solver status %d, initial boxes legal %d\n", Na, Nt,
                  R.t_device * 1e3, R.t_total * 1e3, (int)R.solver_status, (int)R.initial_static_legal);
      for (int a = 0; a < Na; ++a)
        std::printf("  agent %d: %d planes, SQP iterations %d, ADMM iterations %d, last OSQP status %d, %.3f ms on its workgroup\n", a,
                    (int)(off[a + 1] - off[a]), num_iterations[a], (int)admm[a], (int)last[a], agent_s[a] * 1e3);
    }
    if (logger_level >= 3)
      for (int a = 0; a < Na; ++a) std::printf("iteration numbers: %d\n", num_iterations[a]);   // (:259, per agent)
    solve_status = R.solver_status;
    initial_static_legal = R.initial_static_legal != 0;
    max_individual_opt_runtime = R.t_max_individual;
  }

  int getSolverStatus() const { return solve_status; }
  double getMaxOfRuntimes() const { return max_individual_opt_runtime; }
  bool get_initial_static_legal() const { return initial_static_legal; }

  std::vector<int> num_iterations;
  std::vector<std::vector<Corridor>> corridors;
  std::vector<int> admm_iterations;  // new: needed for the agent-QP-iterations/s metric

 private:
  int solve_status = 0;
  double max_individual_opt_runtime = -1;
  bool initial_static_legal = true;
};

// The whole DO phase of csdo.cc:111-159 on the reference's own containers: InterpolateInitalGuess, findNeighborPairsByTrustRegion,
// calcEqualInterPlanes and the SolverDSQP constructor in ONE library call (csdo_do_phase; include/csdo_dsqp.h).  Same getters and
// public members as SolverDSQP, plus the bridge's verdict on the initial guess.
//   PathT: `states` = sequence of pair<State, cost> with State fields x, y, yaw; `actions` = sequence of pair<Action, cost>, Action
//          convertible to int (libMultiRobotPlanning::PlanResult<State, Action, double>, hybrid_a_star/planresult.h:30-44);
//   GoalRange: sequence of states with x, y, yaw (Instance::goal_states); the rest as for SolverDSQP.
// What it does not hand back is x0_bar (csdo.cc dumps it with --initial_guess and with the corridors): csdo_preprocess gives that.
class DoPhase {
 public:
  template <class OptRes, class PathT, class GoalRange, class ObstacleRange, class Parm>
  DoPhase(std::vector<std::vector<OptRes>>& solutions, const std::vector<PathT>& coarse_paths, const GoalRange& goals, double dimx,
          double dimy, const ObstacleRange& obstacles, const Parm& param, int logger_level = 2, int device = 0,
          const csdo_vehicle* vehicle = nullptr, int solve_refinement = 0) {
    const int Na = (int)coarse_paths.size();
    std::vector<double> states, goal_xyz, obs;
    std::vector<int32_t> actions, path_off{0};
    for (const PathT& p : coarse_paths) {
      for (const auto& s : p.states) {
        states.push_back(s.first.x);
        states.push_back(s.first.y);
        states.push_back(s.first.yaw);
      }
      for (const auto& a : p.actions) actions.push_back((int32_t)a.first);
      path_off.push_back((int32_t)(states.size() / 3));
    }
    for (const auto& g : goals) {
      goal_xyz.push_back(g.x);
      goal_xyz.push_back(g.y);
      goal_xyz.push_back(g.yaw);
    }
    for (const auto& o : obstacles) {
      obs.push_back(o.x);
      obs.push_back(o.y);
      obs.push_back(o.r);
    }
    if ((int)goal_xyz.size() != 3 * Na || Na < 1) throw std::runtime_error("csdo::DoPhase: one goal per path expected");
    csdo_vehicle veh;
    if (vehicle) veh = *vehicle; else csdo_vehicle_default(&veh);
    csdo_qp_parm parm;
    csdo_qp_parm_default(&veh, &parm);
    parm.r_trust = param.r_trust;
    parm.max_omega = param.max_omega;
    parm.max_v = param.max_v;
    parm.max_iter = (double)param.max_iter;
    parm.delta_solution_threshold = param.delta_solution_threshold;
    parm.max_violation = param.max_violation;
    parm.osqp_max_iter = param.osqp_max_iter;
    parm.num_interpolation = param.num_interpolation;
    parm.dt = param.dt;
    parm.fixed_corridor = param.fixed_corridor ? 1 : 0;
    parm.solve_refinement = solve_refinement;
    csdo_coarse_world W{};
    W.states = states.data();
    W.actions = actions.data();
    W.path_off = path_off.data();
    W.goals = goal_xyz.data();
    W.obstacles = obs.data();
    W.Na = Na;
    W.n_obs = (int32_t)(obs.size() / 3);
    W.dimx = dimx;
    W.dimy = dimy;
    const int Nt = csdo_do_phase_horizon(path_off.data(), Na, &parm);
    if (Nt < 2) throw std::runtime_error("csdo::DoPhase: paths of fewer than two states");
    std::vector<double> sol((size_t)Na * Nt * 6), cor((size_t)Na * Nt * 8);
    std::vector<int32_t> admm(Na), last(Na);
    num_iterations.assign(Na, 0);
    csdo_result R{};
    R.solutions = sol.data();
    R.corridors = cor.data();
    R.sqp_iters = num_iterations.data();
    R.admm_iters = admm.data();
    R.last_status = last.data();
    int32_t inter_legal = 1;
    csdo_handle h = nullptr;
    int rc = csdo_dsqp_create(&h, device);
    if (rc == CSDO_OK) rc = csdo_do_phase(h, &W, 1, &veh, &parm, &R, &inter_legal, nullptr);
    if (h) csdo_dsqp_destroy(h);
    if (rc != CSDO_OK) throw std::runtime_error("csdo_do_phase failed with code " + std::to_string(rc));
    solutions.assign(Na, std::vector<OptRes>(Nt));
    corridors.assign(Na, std::vector<Corridor>(Nt));
    for (int a = 0; a < Na; ++a)
      for (int t = 0; t < Nt; ++t) {
        const double* s = &sol[((size_t)a * Nt + t) * 6];
        OptRes& r = solutions[a][t];
        r.x = s[0]; r.y = s[1]; r.yaw = s[2]; r.steer = s[3]; r.v = s[4]; r.d_steer = s[5];
        const double* c = &cor[((size_t)a * Nt + t) * 8];
        corridors[a][t] = Corridor{c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7]};
      }
    admm_iterations.assign(admm.begin(), admm.end());
    if (logger_level >= 2)
      std::printf("csdo::DoPhase: %d agents, Nt = %d, kernels %.3f ms, call %.3f ms, solver status %d, initial guess legal %d / %d\n", Na, Nt,
                  R.t_device * 1e3, R.t_total * 1e3, (int)R.solver_status, (int)inter_legal, (int)R.initial_static_legal);
    solve_status = R.solver_status;
    initial_static_legal = R.initial_static_legal != 0;
    initial_inter_legal = inter_legal != 0;
    max_individual_opt_runtime = R.t_max_individual;
  }

  int getSolverStatus() const { return solve_status; }
  double getMaxOfRuntimes() const { return max_individual_opt_runtime; }
  bool get_initial_static_legal() const { return initial_static_legal; }
  bool get_initial_inter_legal() const { return initial_inter_legal; }   // findNeighborPairsByTrustRegion's return value (csdo.cc:120-126)

  std::vector<int> num_iterations;
  std::vector<std::vector<Corridor>> corridors;
  std::vector<int> admm_iterations;

 private:
  int solve_status = 0;
  double max_individual_opt_runtime = -1;
  bool initial_static_legal = true, initial_inter_legal = true;
};

}  // namespace csdo
