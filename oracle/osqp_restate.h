// ORACLE — TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product path;
// only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may build, load or call it.
//
// CPU restatement of the QP solver the reference hands its per-agent QPs to:
//   OSQP 0.6.3 (README.md:16 of the reference; call sites sqp/dsqp_solver.cc:457-502,549)
//   with its bundled QDLDL LDL^T (lin_sys/direct/qdldl).
// Neither library is in /root/reference nor installed in the build container, so this file restates the
// published algorithm of OSQP v0.6.x (src/osqp.c, auxil.c, scaling.c, lin_alg.c, kkt.c, qdldl.c).
// PARITY UNPINNED: the reference ships no golden vectors and cannot be built here, so this restatement is
// checked against an independent dense KKT/active-set solve (tests/test_oracle_qp.py), not against OSQP itself.
//
// Deliberate, documented differences from upstream:
//  * adaptive_rho_interval: upstream default 0 picks the interval from wall-clock timing (osqp.c, PROFILING
//    branch).  Here it is a fixed parameter (default 25 = the smallest value the upstream rule can yield).
//  * fill-reducing ordering: upstream uses AMD; here constraint rows are eliminated first and variables follow
//    in time-major order, which is what a minimum-degree ordering does on this banded KKT (rounding-level effect).
#pragma once
#include <cstdint>
#include <vector>

namespace csdo_oracle {

// Compressed-sparse-column matrix, row indices ascending inside a column (what Eigen::makeCompressed hands to
// csc_matrix() at sqp/dsqp_solver.cc:437-468).
struct Csc {
  int m = 0, n = 0;
  std::vector<int> p;     // n+1 column pointers
  std::vector<int> i;     // row indices
  std::vector<double> x;  // values
  int nnz() const { return p.empty() ? 0 : p.back(); }
};

// Triplet builder -> CSC with sorted rows (duplicates are an error in Eigen::insert; we assert none).
struct TripletList {
  std::vector<int> r, c;
  std::vector<double> v;
  void add(int row, int col, double val) { r.push_back(row); c.push_back(col); v.push_back(val); }
};
Csc csc_from_triplets(int m, int n, const TripletList& t);

// OSQP constants (include/constants.h, v0.6.x)
constexpr double OSQP_INFTY = 1e30;
constexpr double RHO_MIN = 1e-6, RHO_MAX = 1e6, RHO_TOL = 1e-4, RHO_EQ_OVER_RHO_INEQ = 1e3;
constexpr double MIN_SCALING = 1e-4, MAX_SCALING = 1e4;
enum Status : int {
  DUAL_INFEASIBLE_INACCURATE = 4,
  PRIMAL_INFEASIBLE_INACCURATE = 3,
  SOLVED_INACCURATE = 2,
  SOLVED = 1,
  MAX_ITER_REACHED = -2,
  PRIMAL_INFEASIBLE = -3,
  DUAL_INFEASIBLE = -4,
  NON_CVX = -7,
  UNSOLVED = -10
};

struct Settings {  // osqp_set_default_settings(), then max_iter overridden (sqp/dsqp_solver.cc:480-487)
  double rho = 0.1, sigma = 1e-6, alpha = 1.6;
  double eps_abs = 1e-3, eps_rel = 1e-3, eps_prim_inf = 1e-4, eps_dual_inf = 1e-4;
  int scaling = 10;
  int max_iter = 4000;
  int check_termination = 25;
  int adaptive_rho = 1;
  int adaptive_rho_interval = 25;  // see header note
  double adaptive_rho_tolerance = 5.0;
};

struct Info {
  int status = UNSOLVED;
  int iter = 0;
  int rho_updates = 0;
  double pri_res = 0, dua_res = 0;
  double rho_final = 0;
};

// Optional per-solve trace used by tests to compare intermediate quantities with the HIP path.
struct Trace {
  std::vector<double> D, E;          // final Ruiz scalings
  double c = 1.0;
  std::vector<double> l_s, u_s;      // scaled bounds
  std::vector<double> rho_vec;       // at exit
  std::vector<double> x_scaled, y_scaled, z_scaled;
  std::vector<double> pri_res_hist, dua_res_hist, rho_hist;  // one entry per termination check
};

// Solve  min 1/2 x'Px + q'x  s.t. l <= Ax <= u  the way osqp_setup / osqp_warm_start_x / osqp_solve do.
// P_triu: upper triangle only.  x_warm: unscaled warm start (y starts at 0, z = A x).  x_out: unscaled primal
// solution (NaN-free only when has_solution(status)); y_out (optional): unscaled dual.
Info osqp_solve_restated(const Csc& P_triu, const std::vector<double>& q, const Csc& A,
                         const std::vector<double>& l, const std::vector<double>& u,
                         const std::vector<double>& x_warm, const Settings& st,
                         std::vector<double>& x_out, std::vector<double>* y_out = nullptr,
                         Trace* trace = nullptr, const std::vector<int>* var_order = nullptr);

inline bool has_solution(int status) {
  return status != PRIMAL_INFEASIBLE && status != PRIMAL_INFEASIBLE_INACCURATE &&
         status != DUAL_INFEASIBLE && status != DUAL_INFEASIBLE_INACCURATE && status != NON_CVX;
}

// Count of nonzeros in the LDL^T factor of the last solve on this thread (reported next to the CPU baseline).
long last_factor_nnz();

}  // namespace csdo_oracle
