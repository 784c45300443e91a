// Host-side packing of csdo_problem worlds into the flat arrays the device program reads
// (csdo_device_types.h).  Pure C++, no HIP: used by capi.hip for the real launch and by the lane-serial test build.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/csdo_dsqp.h"
#include "csdo_device_types.h"

namespace csdo {

struct HostBatch {
  std::vector<AgentDesc> agents;
  std::vector<WorldDesc> worlds;
  std::vector<double> x0;
  std::vector<PlaneDev> planes;
  std::vector<int32_t> tstart;
  std::vector<double> obstacles;
  std::vector<int32_t> world_first_agent;  // [n_worlds+1]
  std::vector<float> est_work;             // [agents] relative work estimate used to order the launch (see pack_worlds)
  int64_t rows_total = 0;      // inter rows (4 per plane)
  int64_t fac_total = 0;       // doubles of factor workspace
  int64_t steps_total = 0;     // sum of Nt over agents
  int max_nt = 0, max_obs = 0, max_planes = 0;
  SolverParams prm{};
};

inline int fac_stride(int Nt) { return (Nt + 1) & ~1; }

inline SolverParams make_params(const csdo_vehicle& v, const csdo_qp_parm& p) {
  SolverParams s{};
  s.f2x = v.f2x;
  s.r2x = v.r2x;
  s.rv = v.rv;
  s.WB = v.WB;
  s.r_turn = v.r;
  s.r_trust = p.r_trust;
  s.max_omega = p.max_omega;
  s.max_v = p.max_v;
  s.delta_solution_threshold = p.delta_solution_threshold;
  s.dt = p.dt;
  s.max_iter = (int32_t)p.max_iter;
  s.osqp_max_iter = p.osqp_max_iter;
  s.fixed_corridor = p.fixed_corridor;
  s.adaptive_rho_interval = p.adaptive_rho_interval > 0 ? p.adaptive_rho_interval : 25;
  // osqp_set_default_settings (OSQP 0.6.3), as used at sqp/dsqp_solver.cc:480-487
  s.rho0 = 0.1;
  s.sigma = 1e-6;
  s.alpha = 1.6;
  s.eps_abs = 1e-3;
  s.eps_rel = 1e-3;
  s.eps_prim_inf = 1e-4;
  s.scaling_passes = 10;
  s.check_termination = 25;
  s.adaptive_rho_tolerance = 5.0;
  return s;
}

// returns CSDO_OK or an error code
inline int pack_worlds(const csdo_problem* worlds, int n_worlds, HostBatch& hb) {
  if (!worlds || n_worlds < 1) return CSDO_EINVAL;
  hb = HostBatch{};
  hb.prm = make_params(worlds[0].veh, worlds[0].parm);
  hb.world_first_agent.push_back(0);
  for (int w = 0; w < n_worlds; ++w) {
    const csdo_problem& W = worlds[w];
    if (W.Na < 1 || W.Nt < 2 || W.n_obs < 0 || !W.x0_bar || !W.plane_off || (W.n_obs > 0 && !W.obstacles))
      return CSDO_EINVAL;
    if (W.Nt > CSDO_MAX_NT) return CSDO_ELIMIT;
    if (W.plane_off[0] != 0) return CSDO_EINVAL;
    for (int a = 0; a < W.Na; ++a)
      if (W.plane_off[a + 1] < W.plane_off[a]) return CSDO_EINVAL;   // CSR offsets must be monotone
    if (W.plane_off[W.Na] > 0 && !W.planes) return CSDO_EINVAL;
    // One launch reads ONE parameter block (vehicle geometry + QpParm): every world of a batch must carry the values of
    // worlds[0] (compared after the same normalisation make_params applies), otherwise CSDO_EINVAL - never a silent
    // solve with the wrong dt / trust radius / iteration caps.
    if (w > 0) {
      const SolverParams pw = make_params(W.veh, W.parm);
      if (std::memcmp(&pw, &hb.prm, sizeof(SolverParams)) != 0) return CSDO_EINVAL;
    }
    WorldDesc wd{};
    wd.dimx = W.dimx;
    wd.dimy = W.dimy;
    wd.obs_off = (int32_t)(hb.obstacles.size() / 3);
    wd.n_obs = W.n_obs;
    hb.obstacles.insert(hb.obstacles.end(), W.obstacles, W.obstacles + (size_t)3 * W.n_obs);
    hb.worlds.push_back(wd);
    hb.max_obs = std::max(hb.max_obs, (int)W.n_obs);
    hb.max_nt = std::max(hb.max_nt, (int)W.Nt);
    for (int a = 0; a < W.Na; ++a) {
      AgentDesc ad{};
      ad.Nt = W.Nt;
      ad.world = w;
      const int k0 = W.plane_off[a], k1 = W.plane_off[a + 1];
      ad.n_planes = k1 - k0;
      hb.max_planes = std::max(hb.max_planes, (int)ad.n_planes);
      ad.x0_off = (int64_t)hb.x0.size();
      hb.x0.insert(hb.x0.end(), W.x0_bar + (size_t)a * W.Nt * 6, W.x0_bar + (size_t)(a + 1) * W.Nt * 6);
      ad.plane_off = (int64_t)hb.planes.size();
      // planes sorted by timestep (stable): a no-op for the reference's pair order, which is t-major
      std::vector<int> order(ad.n_planes);
      for (int k = 0; k < ad.n_planes; ++k) order[k] = k0 + k;
      std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return W.planes[p].t < W.planes[q].t; });
      ad.tstart_off = (int64_t)hb.tstart.size();
      std::vector<int32_t> ts(W.Nt + 1, 0);
      for (int k : order) {
        const csdo_plane& pl = W.planes[k];
        if (pl.t < 0 || pl.t >= W.Nt) return CSDO_EINVAL;
        PlaneDev pd{};
        pd.t = pl.t;
        std::memcpy(pd.c, pl.c, sizeof(pd.c));
        hb.planes.push_back(pd);
        ts[pl.t + 1]++;
      }
      for (int t = 0; t < W.Nt; ++t) ts[t + 1] += ts[t];
      hb.tstart.insert(hb.tstart.end(), ts.begin(), ts.end());
      // Work estimate for the launch order.  Agents whose initial guess violates one of its separating planes run
      // their QPs to the iteration cap for most SQP iterations (measured on the benchmark sets: the plane residual at
      // x0_bar separates the ~6 % of agents that take 10x longer from the rest almost perfectly); everybody else
      // converges in 2-3 SQP iterations.  Per-iteration cost grows with the horizon and the plane count.
      double worst = 0.0;
      for (int k = k0; k < k1; ++k) {
        const csdo_plane& pl = W.planes[k];
        if (pl.t < 0 || pl.t >= W.Nt) return CSDO_EINVAL;
        const double* xs = W.x0_bar + ((size_t)a * W.Nt + pl.t) * 6;
        const double cy = std::cos(xs[2]), sy = std::sin(xs[2]);
        for (int r = 0; r < 4; ++r) {   // rows 0,1: front disc centre, rows 2,3: rear disc (sqp/inter_agent_cons.cc:71-140)
          const double off = r < 2 ? hb.prm.f2x : hb.prm.r2x;
          const double res = pl.c[3 * r] * (xs[0] + off * cy) + pl.c[3 * r + 1] * (xs[1] + off * sy) + pl.c[3 * r + 2];
          worst = std::max(worst, res);
        }
      }
      hb.est_work.push_back((float)((worst > 0.0 ? 10.0 : 1.0) * (2.0 * W.Nt + ad.n_planes)));
      ad.rows_off = hb.rows_total;
      hb.rows_total += (int64_t)4 * ad.n_planes;
      ad.fac_off = hb.fac_total;
      hb.fac_total += (int64_t)(FAC_E_DOUBLES + FAC_X_DOUBLES + COLD_DOUBLES) * fac_stride(W.Nt);
      ad.out_off = hb.steps_total;
      hb.steps_total += W.Nt;
      hb.agents.push_back(ad);
    }
    hb.world_first_agent.push_back((int32_t)hb.agents.size());
  }
  return CSDO_OK;
}

// Scatter packed outputs back into per-world csdo_result buffers and aggregate the solver status the way
// SolverDSQP does (sqp/dsqp_solver.cc:1224-1243: start at 2, signed assignment when |s| exceeds it).
inline void unpack_results(const HostBatch& hb, const csdo_problem* worlds, int n_worlds, const double* sol,
                           const double* corr, const int32_t* sqp_iters, const int32_t* admm_iters,
                           const int32_t* last_status, const int32_t* static_legal, csdo_result* results) {
  for (int w = 0; w < n_worlds; ++w) {
    const int a0 = hb.world_first_agent[w], a1 = hb.world_first_agent[w + 1];
    csdo_result& R = results[w];
    bool any_bad = false;
    int worst = 2, legal = 1;
    for (int a = a0; a < a1; ++a) {
      const AgentDesc& ad = hb.agents[a];
      const int la = a - a0;
      std::memcpy(R.solutions + (size_t)la * ad.Nt * 6, sol + ad.out_off * 6, sizeof(double) * ad.Nt * 6);
      std::memcpy(R.corridors + (size_t)la * ad.Nt * 8, corr + ad.out_off * 8, sizeof(double) * ad.Nt * 8);
      R.sqp_iters[la] = sqp_iters[a];
      R.admm_iters[la] = admm_iters[a];
      R.last_status[la] = last_status[a];
      const int s = last_status[a];
      if (std::abs(s) > 1) {
        any_bad = true;
        if (std::abs(s) > worst) worst = s;
      }
      if (!static_legal[a]) legal = 0;
    }
    R.solver_status = any_bad ? worst : 1;
    R.initial_static_legal = legal;
    (void)worlds;
  }
}

}  // namespace csdo
