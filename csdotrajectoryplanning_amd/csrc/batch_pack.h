// Host-side packing of csdo_problem worlds into the flat arrays the device program reads
// (csdo_device_types.h).  Pure C++, no HIP: used by capi.hip for the real launch and by the lane-serial test build.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <functional>
#include <thread>
#include <vector>

#include "../../include/csdo_dsqp.h"
#include "csdo_device_types.h"
#include "host_pool.h"

namespace csdo {

struct HostBatch {
  std::vector<AgentDesc> agents;
  std::vector<WorldDesc> worlds;
  std::vector<double> x0;
  std::vector<PlaneDev> planes;
  std::vector<int32_t> tstart;
  std::vector<double> obstacles;
  std::vector<int32_t> world_first_agent;  // [n_worlds+1]
  std::vector<float> est_work;             // [agents] relative work estimate: CU shares of the launch groups (see pack_worlds)
  std::vector<float> launch_rank;          // [agents] likelihood of being one of the long runners: launch order inside a group
  int64_t rows_total = 0;      // inter rows (4 per plane)
  int64_t fac_total = 0;       // doubles of factor workspace
  int64_t steps_total = 0;     // sum of Nt over agents
  int max_nt = 0, max_obs = 0, max_planes = 0;
  SolverParams prm{};
};

inline int fac_stride(int Nt) { return (Nt + 1) & ~1; }

// The four big arrays of a packed batch (x0, planes, tstart, obstacles: 30 MB for 2000 vehicles) live in the batch's own vectors, or
// where the caller of pack_worlds has them put: capi.hip hands out slots of its page-locked staging arena, so that nothing is zeroed,
// page-faulted in or copied a second time per upload.  The placer is called once, with the element counts, before anything is filled.
struct PackSizes { size_t agents, worlds, x0, planes, tstart, obstacles; };
struct PackPlace { double* x0; PlaneDev* planes; int32_t* tstart; double* obstacles; };
using PackPlacer = std::function<int(const PackSizes&, PackPlace&)>;
// a placer that places nothing: descriptors, offsets and the work estimates only (csdo_dsqp_estimate_work, the sharding pass of a
// multi-device upload) - the big arrays are neither allocated nor filled
inline int pack_nothing(const PackSizes&, PackPlace& at) {
  at = PackPlace{nullptr, nullptr, nullptr, nullptr};
  return CSDO_OK;
}

inline SolverParams make_params(const csdo_vehicle& v, const csdo_qp_parm& p) {
  SolverParams s{};
  s.f2x = v.f2x;
  s.r2x = v.r2x;
  s.rv = v.rv;
  s.WB = v.WB;
  s.r_turn = v.r;
  s.steer_max = std::atan(v.WB / v.r);
  s.r_trust = p.r_trust;
  s.max_omega = p.max_omega;
  s.max_v = p.max_v;
  s.delta_solution_threshold = p.delta_solution_threshold;
  s.dt = p.dt;
  s.max_iter = (int32_t)p.max_iter;
  s.osqp_max_iter = p.osqp_max_iter;
  s.fixed_corridor = p.fixed_corridor;
  s.adaptive_rho_interval = p.adaptive_rho_interval > 0 ? p.adaptive_rho_interval : 25;
  s.solve_refinement = (p.solve_refinement == 1 || p.solve_refinement == 2) ? p.solve_refinement : 0;   // 1 full, 2 lagged
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

// returns CSDO_OK or an error code.  Two passes: sizes and offsets of every world (serial, cheap), then blocks of agents are
// filled in parallel by the library's host threads (the per-agent plane ordering and the work estimate - a cos / sin per plane -
// are what takes time: 41 ms single-threaded for the 3000-agent batch).
inline int pack_worlds(const csdo_problem* worlds, int n_worlds, HostBatch& hb, const PackPlacer* placer = nullptr) {
  if (!worlds || n_worlds < 1) return CSDO_EINVAL;
  hb = HostBatch{};
  hb.prm = make_params(worlds[0].veh, worlds[0].parm);
  struct Off { size_t agent, x0, plane, tstart, obs; int64_t rows, fac, steps; };
  std::vector<Off> off(n_worlds + 1);
  Off cur{0, 0, 0, 0, 0, 0, 0, 0};
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
    off[w] = cur;
    const int64_t K = W.plane_off[W.Na];
    cur.agent += (size_t)W.Na;
    cur.x0 += (size_t)W.Na * W.Nt * 6;
    cur.plane += (size_t)K;
    cur.tstart += (size_t)W.Na * (W.Nt + 1);
    cur.obs += (size_t)3 * W.n_obs;
    cur.rows += 4 * K;
    cur.fac += (int64_t)W.Na * (FAC_E_DOUBLES + FAC_X_DOUBLES + COLD_DOUBLES) * fac_stride(W.Nt);
    cur.steps += (int64_t)W.Na * W.Nt;
    hb.world_first_agent.push_back((int32_t)cur.agent);
    hb.max_obs = std::max(hb.max_obs, (int)W.n_obs);
    hb.max_nt = std::max(hb.max_nt, (int)W.Nt);
  }
  off[n_worlds] = cur;
  hb.agents.resize(cur.agent);
  hb.est_work.resize(cur.agent);
  hb.launch_rank.resize(cur.agent);
  hb.worlds.resize(n_worlds);
  PackPlace at{nullptr, nullptr, nullptr, nullptr};
  if (placer) {
    const int prc_ = (*placer)(PackSizes{cur.agent, (size_t)n_worlds, cur.x0, cur.plane, cur.tstart, cur.obs}, at);
    if (prc_ != CSDO_OK) return prc_;
  } else {
    hb.x0.resize(cur.x0);
    hb.planes.resize(cur.plane);
    hb.tstart.resize(cur.tstart);
    hb.obstacles.resize(cur.obs);
    at = PackPlace{hb.x0.data(), hb.planes.data(), hb.tstart.data(), hb.obstacles.data()};
  }
  hb.rows_total = cur.rows;
  hb.fac_total = cur.fac;
  hb.steps_total = cur.steps;
  std::atomic<int> err{CSDO_OK}, max_planes{0};
  // the unit of host work is a block of agents of one world (a streamed DO phase's first chunk is five worlds: per world, it would
  // be five threads' work whatever the machine has)
  constexpr int AGENT_BLOCK = 8;
  std::vector<std::pair<int32_t, int32_t>> items;
  for (int w = 0; w < n_worlds; ++w)
    for (int a0 = 0; a0 < worlds[w].Na; a0 += AGENT_BLOCK) items.emplace_back(w, a0);
  auto fill = [&](const int item) {
    {
      const int w = items[(size_t)item].first, a_lo = items[(size_t)item].second;
      const csdo_problem& W = worlds[w];
      const int a_hi = std::min(a_lo + AGENT_BLOCK, (int)W.Na);
      const Off& o = off[w];
      if (a_lo == 0) {
        WorldDesc wd{};
        wd.dimx = W.dimx;
        wd.dimy = W.dimy;
        wd.obs_off = (int32_t)(o.obs / 3);
        wd.n_obs = W.n_obs;
        hb.worlds[w] = wd;
        if (W.n_obs && at.obstacles) std::memcpy(at.obstacles + o.obs, W.obstacles, sizeof(double) * 3 * (size_t)W.n_obs);
      }
      if (at.x0)
        std::memcpy(at.x0 + o.x0 + (size_t)a_lo * W.Nt * 6, W.x0_bar + (size_t)a_lo * W.Nt * 6,
                    sizeof(double) * (size_t)(a_hi - a_lo) * W.Nt * 6);
      const int64_t fac_per_agent = (int64_t)(FAC_E_DOUBLES + FAC_X_DOUBLES + COLD_DOUBLES) * fac_stride(W.Nt);
      std::vector<int> order, near_obs;
      std::vector<int32_t> ts((size_t)W.Nt + 1);
      for (int a = a_lo; a < a_hi; ++a) {
        AgentDesc ad{};
        ad.Nt = W.Nt;
        ad.world = w;
        const int k0 = W.plane_off[a], k1 = W.plane_off[a + 1];
        ad.n_planes = k1 - k0;
        int mp = max_planes.load();
        while (ad.n_planes > mp && !max_planes.compare_exchange_weak(mp, ad.n_planes)) {}
        ad.x0_off = (int64_t)(o.x0 + (size_t)a * W.Nt * 6);
        ad.plane_off = (int64_t)(o.plane + (size_t)k0);
        // planes sorted by timestep (stable): a no-op for the reference's pair order, which is t-major
        order.resize(ad.n_planes);
        for (int k = 0; k < ad.n_planes; ++k) order[k] = k0 + k;
        std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return W.planes[p].t < W.planes[q].t; });
        ad.tstart_off = (int64_t)(o.tstart + (size_t)a * (W.Nt + 1));
        std::fill(ts.begin(), ts.end(), 0);
        PlaneDev* dst = at.planes ? at.planes + ad.plane_off : nullptr;
        // Launch order.  A few per cent of the agents take 3-6x the median time (QPs that run to the iteration cap for
        // most SQP iterations) and decide the makespan of a batch unless they start early.  Measured on the two benchmark
        // sets with this repository's front-end paths (scripts/agent_times.py, 4500 agents): what marks them is a tight
        // spot in the initial guess - timesteps whose disc centres are within 0.5 m of an inflated obstacle, and
        // separating planes the guess violates; horizon, plane count and the worst plane residual alone do not (list
        // scheduling by them is worse than a random order).  Ordering by the fraction of the horizon spent in tight spots
        // brings the simulated makespan of the map100 set from 157 ms to 117 ms (110 ms with the true times known).
        int n_violated = 0;
        for (int k = 0; k < ad.n_planes; ++k) {
          const csdo_plane& pl = W.planes[order[k]];
          if (pl.t < 0 || pl.t >= W.Nt) {
            err.store(CSDO_EINVAL);
            return;
          }
          PlaneDev pd{};
          pd.t = pl.t;
          std::memcpy(pd.c, pl.c, sizeof(pd.c));
          if (dst) dst[k] = pd;
          ts[pl.t + 1]++;
          const double* xs = W.x0_bar + ((size_t)a * W.Nt + pl.t) * 6;
          const double cy = std::cos(xs[2]), sy = std::sin(xs[2]);
          double worst = 0.0;
          for (int r = 0; r < 4; ++r) {   // rows 0,1: front disc centre, rows 2,3: rear disc (sqp/inter_agent_cons.cc:71-140)
            const double offx = r < 2 ? hb.prm.f2x : hb.prm.r2x;
            const double res = pl.c[3 * r] * (xs[0] + offx * cy) + pl.c[3 * r + 1] * (xs[1] + offx * sy) + pl.c[3 * r + 2];
            worst = std::max(worst, res);
          }
          n_violated += worst > 0.0;
        }
        int n_near = 0;
        // only the obstacles near the path's bounding box can be near one of its timesteps
        near_obs.clear();
        if (W.n_obs > 0) {
          double bx0 = 1e300, bx1 = -1e300, by0 = 1e300, by1 = -1e300;
          for (int t = 0; t < W.Nt; ++t) {
            const double* xs = W.x0_bar + ((size_t)a * W.Nt + t) * 6;
            bx0 = std::min(bx0, xs[0]); bx1 = std::max(bx1, xs[0]);
            by0 = std::min(by0, xs[1]); by1 = std::max(by1, xs[1]);
          }
          const double reach = std::max(std::fabs(hb.prm.f2x), std::fabs(hb.prm.r2x)) + hb.prm.rv + 0.5;
          for (int j = 0; j < W.n_obs; ++j) {
            const double* ob = W.obstacles + 3 * (size_t)j;
            const double m = reach + ob[2];
            if (ob[0] > bx0 - m && ob[0] < bx1 + m && ob[1] > by0 - m && ob[1] < by1 + m) near_obs.push_back(j);
          }
        }
        for (int t = 0; t < W.Nt && !near_obs.empty(); ++t) {
          const double* xs = W.x0_bar + ((size_t)a * W.Nt + t) * 6;
          const double cy = std::cos(xs[2]), sy = std::sin(xs[2]);
          const double fx = xs[0] + hb.prm.f2x * cy, fy = xs[1] + hb.prm.f2x * sy;
          const double rx = xs[0] + hb.prm.r2x * cy, ry = xs[1] + hb.prm.r2x * sy;
          bool near = false;
          for (size_t jj = 0; jj < near_obs.size() && !near; ++jj) {
            const double* ob = W.obstacles + 3 * (size_t)near_obs[jj];
            const double lim = ob[2] + hb.prm.rv + 0.5;
            const double df = (fx - ob[0]) * (fx - ob[0]) + (fy - ob[1]) * (fy - ob[1]);
            const double dr = (rx - ob[0]) * (rx - ob[0]) + (ry - ob[1]) * (ry - ob[1]);
            near = std::min(df, dr) < lim * lim;
          }
          n_near += near;
        }
        for (int t = 0; t < W.Nt; ++t) ts[t + 1] += ts[t];
        if (at.tstart) std::memcpy(at.tstart + ad.tstart_off, ts.data(), sizeof(int32_t) * ts.size());
        const double tight = (n_near + 0.2 * n_violated) / (double)W.Nt;
        hb.launch_rank[o.agent + a] = (float)tight;
        hb.est_work[o.agent + a] = (float)((1.0 + 10.0 * std::min(tight, 1.0)) * (2.0 * W.Nt + ad.n_planes));
        ad.rows_off = o.rows + (int64_t)4 * k0;
        ad.fac_off = o.fac + (int64_t)a * fac_per_agent;
        ad.out_off = o.steps + (int64_t)a * W.Nt;
        hb.agents[o.agent + a] = ad;
      }
    }
  };
  const int prc = parallel_for((int)items.size(), 64, fill);
  hb.max_planes = max_planes.load();
  return err.load() != CSDO_OK ? err.load() : prc;
}

// Scatter packed outputs back into per-world csdo_result buffers and aggregate the solver status the way
// SolverDSQP does (sqp/dsqp_solver.cc:1224-1243: start at 2, signed assignment when |s| exceeds it).
inline void unpack_results(const HostBatch& hb, const csdo_problem* worlds, int n_worlds, const double* sol,
                           const double* corr, const int32_t* sqp_iters, const int32_t* admm_iters,
                           const int32_t* last_status, const int32_t* static_legal, csdo_result* results) {
  // blocks of agents scattered by the host threads (57 MB for the 3000-agent batch); the worlds' aggregates afterwards
  constexpr int AGENT_BLOCK = 16;
  std::vector<std::pair<int32_t, int32_t>> items;
  for (int w = 0; w < n_worlds; ++w)
    for (int a = hb.world_first_agent[w]; a < hb.world_first_agent[w + 1]; a += AGENT_BLOCK) items.emplace_back(w, a);
  auto scatter = [&](const int item) {
    const int w = items[(size_t)item].first, a0 = hb.world_first_agent[w];
    const int a_lo = items[(size_t)item].second, a_hi = std::min(a_lo + AGENT_BLOCK, (int)hb.world_first_agent[w + 1]);
    csdo_result& R = results[w];
    for (int a = a_lo; a < a_hi; ++a) {
      const AgentDesc& ad = hb.agents[a];
      const int la = a - a0;
      std::memcpy(R.solutions + (size_t)la * ad.Nt * 6, sol + ad.out_off * 6, sizeof(double) * ad.Nt * 6);
      std::memcpy(R.corridors + (size_t)la * ad.Nt * 8, corr + ad.out_off * 8, sizeof(double) * ad.Nt * 8);
      R.sqp_iters[la] = sqp_iters[a];
      R.admm_iters[la] = admm_iters[a];
      R.last_status[la] = last_status[a];
    }
  };
  (void)parallel_for((int)items.size(), 64, scatter);
  for (int w = 0; w < n_worlds; ++w) {
    bool any_bad = false;
    int worst = 2, legal = 1;
    for (int a = hb.world_first_agent[w]; a < hb.world_first_agent[w + 1]; ++a) {
      const int s = last_status[a];
      if (std::abs(s) > 1) {
        any_bad = true;
        if (std::abs(s) > worst) worst = s;
      }
      if (!static_legal[a]) legal = 0;
    }
    results[w].solver_status = any_bad ? worst : 1;
    results[w].initial_static_legal = legal;
  }
  (void)worlds;
}

}  // namespace csdo
