// TEST: the C++ mirror csdo::SolverDSQP (csdotrajectoryplanning_amd/host/solver_dsqp.hpp) compiled against stand-in types
// shaped like the reference's own (libMultiRobotPlanning::OptimizeResult / QpParm, sqp/common.h:14-52; InterPlane,
// sqp/inter_agent_cons.h:47-63; Location in an unordered_set, common/motion_planning.h:79-106) - the containers csdo.cc
// passes at csdo.cc:146-147 - with no conversion.  Reads a world from a flat binary file written by the Python test,
// solves it through the mirror, writes the results back; tests/test_cpp_mirror.py compares them with the ctypes path.
//   usage: mirror_main <in.bin> <out.bin>
#include <string>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <unordered_set>
#include <vector>

#include "../../csdotrajectoryplanning_amd/host/solver_dsqp.hpp"

namespace libMultiRobotPlanning {   // field order and types as in the reference headers
struct OptimizeResult {
  double x, y, yaw, v, a, steer, d_steer;
};
struct QpParm {
  double r_trust, max_omega, max_v, max_iter, delta_solution_threshold, max_violation;
  int osqp_max_iter;
  double dt;
  int num_interpolation;
  bool fixed_corridor;
};
struct Location {
  Location(double x, double y, double r = 0.8) : x(x), y(y), r(r) {}
  double x, y, r;
  bool operator==(const Location& o) const { return x == o.x && y == o.y; }
};
}  // namespace libMultiRobotPlanning
namespace std {
template <>
struct hash<libMultiRobotPlanning::Location> {
  size_t operator()(const libMultiRobotPlanning::Location& s) const {
    return std::hash<double>()(s.x) * 31 + std::hash<double>()(s.y);
  }
};
}  // namespace std
struct InterPlane {   // global namespace in the reference (sqp/inter_agent_cons.h:47)
  int t;
  double a_f2f, b_f2f, c_f2f, a_f2r, b_f2r, c_f2r, a_r2f, b_r2f, c_r2f, a_r2r, b_r2r, c_r2r;
};

using namespace libMultiRobotPlanning;

static bool rd(FILE* f, void* p, size_t n) { return std::fread(p, 1, n, f) == n; }

int main(int argc, char** argv) {
  if (argc < 3 || argc > 6) return 2;   // in out [devices|-] [refine] [log]
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int32_t hdr[4];   // Na, Nt, n_obs, n_planes
  double dims[2];
  QpParm param{};
  double parm_d[7];   // r_trust, max_omega, max_v, max_iter, delta_solution_threshold, max_violation, dt
  int32_t parm_i[3];  // osqp_max_iter, num_interpolation, fixed_corridor
  if (!rd(f, hdr, sizeof hdr) || !rd(f, dims, sizeof dims) || !rd(f, parm_d, sizeof parm_d) || !rd(f, parm_i, sizeof parm_i))
    return 4;
  const int Na = hdr[0], Nt = hdr[1], n_obs = hdr[2];
  param.r_trust = parm_d[0]; param.max_omega = parm_d[1]; param.max_v = parm_d[2]; param.max_iter = parm_d[3];
  param.delta_solution_threshold = parm_d[4]; param.max_violation = parm_d[5]; param.dt = parm_d[6];
  param.osqp_max_iter = parm_i[0]; param.num_interpolation = parm_i[1]; param.fixed_corridor = parm_i[2] != 0;
  std::vector<std::vector<OptimizeResult>> x0_bar(Na, std::vector<OptimizeResult>(Nt)), optimize_res;
  for (int a = 0; a < Na; ++a)
    for (int t = 0; t < Nt; ++t) {
      double g[6];
      if (!rd(f, g, sizeof g)) return 4;
      x0_bar[a][t] = OptimizeResult{g[0], g[1], g[2], g[4], 0.0, g[3], g[5]};
    }
  std::vector<int32_t> off(Na + 1);
  if (!rd(f, off.data(), sizeof(int32_t) * (Na + 1))) return 4;
  std::vector<std::vector<InterPlane>> inter_planes(Na);
  for (int a = 0; a < Na; ++a)
    for (int k = off[a]; k < off[a + 1]; ++k) {
      int32_t t;
      double c[12];
      if (!rd(f, &t, sizeof t) || !rd(f, c, sizeof c)) return 4;
      inter_planes[a].push_back(InterPlane{t, c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[7], c[8], c[9], c[10], c[11]});
    }
  // an unordered_set iterates in bucket order; the obstacles of the test world never overlap a disc centre twice, so the
  // order does not matter
  std::unordered_set<Location> obstacles;
  for (int k = 0; k < n_obs; ++k) {
    double o[3];
    if (!rd(f, o, sizeof o)) return 4;
    obstacles.insert(Location(o[0], o[1], o[2]));
  }
  std::fclose(f);

  int rc_status = 0;
  try {
    // optional third argument: GPU ordinals "0,0" - the same constructor over several devices (csdo_dsqp_create_multi)
    std::vector<int> devices;
    if (argc > 3 && argv[3][0] != '-')
      for (const char* p = argv[3]; *p;) {
        devices.push_back((int)std::strtol(p, const_cast<char**>(&p), 10));
        if (*p == ',') ++p;
      }
    // optional fourth argument "refine": csdo_qp_parm::solve_refinement through the constructor's last parameter
    const bool refine = argc > 4 && std::string(argv[4]) == "refine";
    const int logger_level = (argc > 5 && std::string(argv[5]) == "log") ? 3 : 0;   // fifth argument "log": the per-agent diagnostics
    csdo::SolverDSQP solver(optimize_res, x0_bar, inter_planes, dims[0], dims[1], obstacles, param, logger_level,
                            /*device*/ 0, /*vehicle*/ nullptr, devices, refine);
    FILE* o = std::fopen(argv[2], "wb");
    if (!o) return 5;
    const int32_t st[2] = {solver.getSolverStatus(), solver.get_initial_static_legal() ? 1 : 0};
    const double tmax = solver.getMaxOfRuntimes();
    std::fwrite(st, sizeof st, 1, o);
    std::fwrite(&tmax, sizeof tmax, 1, o);
    for (int a = 0; a < Na; ++a) {
      const int32_t it[2] = {solver.num_iterations[a], solver.admm_iterations[a]};
      std::fwrite(it, sizeof it, 1, o);
    }
    for (int a = 0; a < Na; ++a)
      for (int t = 0; t < Nt; ++t) {
        const OptimizeResult& r = optimize_res[a][t];
        const double g[6] = {r.x, r.y, r.yaw, r.steer, r.v, r.d_steer};
        std::fwrite(g, sizeof g, 1, o);
        const csdo::Corridor& c = solver.corridors[a][t];
        const double cc[8] = {c.xf_min, c.xf_max, c.yf_min, c.yf_max, c.xr_min, c.xr_max, c.yr_min, c.yr_max};
        std::fwrite(cc, sizeof cc, 1, o);
      }
    std::fclose(o);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "mirror_main: %s\n", e.what());
    rc_status = 1;
  }
  return rc_status;
}
