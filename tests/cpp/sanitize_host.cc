// AddressSanitizer + UndefinedBehaviorSanitizer run of the host code in front of a launch (tests/test_host_pool.py builds it with
// csrc/bridge_host.cc): thirty random worlds through the blocked bridge on the host threads, then packed twice - into the batch's own
// arrays and estimates-only - with the same work estimates.
#include "../../csdotrajectoryplanning_amd/csrc/bridge_host.h"
#include "../../csdotrajectoryplanning_amd/csrc/batch_pack.h"
#include <cstdio>
#include <random>
int main() {
  csdo_vehicle veh{}; veh.r = 3; veh.deltat = 0.706; veh.LF = 2; veh.LB = 1; veh.car_width = 2; veh.WB = 1; veh.f2x = 1.25; veh.r2x = -0.25; veh.rv = 1.25; veh.obs_radius = 0.8;
  csdo_qp_parm parm{}; parm.r_trust = 2; parm.max_omega = 0.07; parm.max_v = 1; parm.max_iter = 10; parm.delta_solution_threshold = 1; parm.max_violation = 1e-3; parm.osqp_max_iter = 400; parm.num_interpolation = 2; parm.dt = 0.88;
  std::mt19937 rng(7);
  std::uniform_real_distribution<double> U(0, 1);
  for (int rep = 0; rep < 30; ++rep) {
    const int Na = 1 + (int)(U(rng) * 40);
    std::vector<double> st, goals; std::vector<int32_t> ac, po{0};
    for (int a = 0; a < Na; ++a) {
      int L = 1 + (int)(U(rng) * 25); if (a == 0 && L < 2) L = 2;
      double x = 30 + 10 * U(rng), y = 30 + 10 * U(rng), yaw = 6.28 * U(rng);
      st.insert(st.end(), {x, y, yaw});
      for (int k = 1; k < L; ++k) { int act = (int)(U(rng) * 7); double s = act == 6 ? 0 : (act < 3 ? 2.1 : -2.1); x += s * std::cos(yaw); y += s * std::sin(yaw); yaw += (act % 3 == 1 ? -0.7 : (act % 3 == 2 ? 0.7 : 0)); st.insert(st.end(), {x, y, yaw}); ac.push_back(act); }
      po.push_back((int32_t)st.size() / 3);
      goals.insert(goals.end(), {x, y, yaw});
    }
    csdo_bridge_out out{};
    int rc = csdo::bridge_preprocess(st.data(), ac.data(), po.data(), Na, goals.data(), &veh, &parm, &out);
    if (rc != 0) { printf("rc %d\n", rc); return 1; }
    // pack it (estimate-only and into vectors)
    csdo_problem P{}; P.Na = out.Na; P.Nt = out.Nt; P.x0_bar = out.x0_bar; P.plane_off = out.plane_off; P.planes = out.planes; P.dimx = P.dimy = 100; P.veh = veh; P.parm = parm;
    csdo::HostBatch hb; 
    if (csdo::pack_worlds(&P, 1, hb) != 0) return 2;
    const csdo::PackPlacer none = csdo::pack_nothing;
    csdo::HostBatch hb2; if (csdo::pack_worlds(&P, 1, hb2, &none) != 0) return 3;
    if (hb.est_work != hb2.est_work) return 4;
    printf("rep %d Na %d Nt %d pairs %d planes %d\n", rep, out.Na, out.Nt, out.n_pairs, out.plane_off[out.Na]);
    csdo::bridge_free(&out);
  }
  puts("ok");
}
