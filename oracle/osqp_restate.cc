// ORACLE — TEST INFRASTRUCTURE ONLY (see osqp_restate.h).
// Restatement of OSQP v0.6.3 + QDLDL for the reference call sequence at sqp/dsqp_solver.cc:457-549:
//   osqp_set_default_settings -> max_iter -> osqp_setup -> osqp_warm_start_x -> osqp_solve -> read x/status.
#include "osqp_restate.h"

#include <algorithm>
#include <cassert>
#include <cmath>
#include <limits>
#include <numeric>
#ifdef CSDO_ORACLE_QUAD
#include <quadmath.h>
#endif

namespace csdo_oracle {

// The arithmetic type of OSQP's linear algebra (scale_data, the LDL^T, the ADMM updates, the residuals and every test on
// them).  double = OSQP as the reference links it.  CSDO_ORACLE_QUAD (oracle/Makefile: libcsdo_oracle_q.so): IEEE binary128 -
// the ARBITER of the chain-parity report (scripts/chain_parity.py): the same algorithm on the same double-precision QP data
// (assembly, safe boxes and the SQP loop stay in double: the reference defines them so) with 113 instead of 53 bits in the
// solve, the solution rounded to double once per QP.  CSDO_ORACLE_LONGDOUBLE: x87 extended (64 bits), a faster, weaker arbiter.
// Constants of the algorithm (sigma, alpha, rho bounds, tolerances) are OSQP's doubles, converted exactly.
#if defined(CSDO_ORACLE_QUAD)
typedef __float128 real;
static inline real r_abs(real a) { return fabsq(a); }
static inline real r_sqrt(real a) { return sqrtq(a); }
#elif defined(CSDO_ORACLE_LONGDOUBLE)
typedef long double real;
static inline real r_abs(real a) { return fabsl(a); }
static inline real r_sqrt(real a) { return sqrtl(a); }
#else
typedef double real;
static inline real r_abs(real a) { return std::fabs(a); }
static inline real r_sqrt(real a) { return std::sqrt(a); }
#endif
static inline real r_max(real a, real b) { return std::max(a, b); }
static inline real r_min(real a, real b) { return std::min(a, b); }
typedef std::vector<real> rvec;

// Csc with values in the solve's arithmetic
struct CscR {
  int m = 0, n = 0;
  std::vector<int> p, i;
  rvec x;
};
static CscR to_real(const Csc& M) {
  CscR R;
  R.m = M.m;
  R.n = M.n;
  R.p = M.p;
  R.i = M.i;
  R.x.assign(M.x.begin(), M.x.end());
  return R;
}
static rvec to_real(const std::vector<double>& v) { return rvec(v.begin(), v.end()); }
struct TripletListR {
  std::vector<int> r, c;
  rvec v;
  void add(int row, int col, real val) { r.push_back(row); c.push_back(col); v.push_back(val); }
};

static thread_local long g_last_factor_nnz = 0;
long last_factor_nnz() { return g_last_factor_nnz; }

template <class M_t, class T_t>
static M_t csc_from_triplets_t(int m, int n, const T_t& t) {
  M_t M;
  M.m = m;
  M.n = n;
  const size_t nz = t.v.size();
  std::vector<size_t> ord(nz);
  std::iota(ord.begin(), ord.end(), size_t(0));
  std::sort(ord.begin(), ord.end(), [&](size_t a, size_t b) {
    if (t.c[a] != t.c[b]) return t.c[a] < t.c[b];
    return t.r[a] < t.r[b];
  });
  M.p.assign(n + 1, 0);
  M.i.resize(nz);
  M.x.resize(nz);
  for (size_t k = 0; k < nz; ++k) {
    const size_t s = ord[k];
    assert(t.r[s] >= 0 && t.r[s] < m && t.c[s] >= 0 && t.c[s] < n);
    if (k > 0) assert(!(t.c[s] == t.c[ord[k - 1]] && t.r[s] == t.r[ord[k - 1]]) && "duplicate entry");
    M.i[k] = t.r[s];
    M.x[k] = t.v[s];
    M.p[t.c[s] + 1]++;
  }
  for (int j = 0; j < n; ++j) M.p[j + 1] += M.p[j];
  return M;
}

Csc csc_from_triplets(int m, int n, const TripletList& t) { return csc_from_triplets_t<Csc, TripletList>(m, n, t); }

// ---------------------------------------------------------------------------------------------------------
// lin_alg.c restated
// ---------------------------------------------------------------------------------------------------------
static real vec_norm_inf(const rvec& v) {
  real mx = 0.0;
  for (real a : v) {
    const real b = r_abs(a);
    if (b > mx) mx = b;
  }
  return mx;
}
static real vec_scaled_norm_inf(const rvec& S, const rvec& v) {
  real mx = 0.0;
  for (size_t k = 0; k < v.size(); ++k) {
    const real b = r_abs(S[k] * v[k]);
    if (b > mx) mx = b;
  }
  return mx;
}
// y (+)= A x, column sweep (mat_vec)
static void mat_vec(const CscR& A, const rvec& x, rvec& y, bool plus_eq) {
  if (!plus_eq) std::fill(y.begin(), y.end(), 0.0);
  for (int j = 0; j < A.n; ++j)
    for (int k = A.p[j]; k < A.p[j + 1]; ++k) y[A.i[k]] += A.x[k] * x[j];
}
// y (+)= A' x (mat_tpose_vec), optionally skipping the diagonal (used to complete P from its upper triangle)
static void mat_tpose_vec(const CscR& A, const rvec& x, rvec& y, bool plus_eq,
                          bool skip_diag) {
  if (!plus_eq) std::fill(y.begin(), y.end(), 0.0);
  for (int j = 0; j < A.n; ++j)
    for (int k = A.p[j]; k < A.p[j + 1]; ++k) {
      const int i = A.i[k];
      if (skip_diag && i == j) continue;
      y[j] += A.x[k] * x[i];
    }
}
static void inf_norm_cols(const CscR& M, rvec& E) {
  std::fill(E.begin(), E.end(), 0.0);
  for (int j = 0; j < M.n; ++j)
    for (int k = M.p[j]; k < M.p[j + 1]; ++k) E[j] = r_max(r_abs(M.x[k]), E[j]);
}
static void inf_norm_rows(const CscR& M, rvec& E) {
  std::fill(E.begin(), E.end(), 0.0);
  for (int j = 0; j < M.n; ++j)
    for (int k = M.p[j]; k < M.p[j + 1]; ++k) E[M.i[k]] = r_max(r_abs(M.x[k]), E[M.i[k]]);
}
static void inf_norm_cols_sym_triu(const CscR& M, rvec& E) {
  std::fill(E.begin(), E.end(), 0.0);
  for (int j = 0; j < M.n; ++j)
    for (int k = M.p[j]; k < M.p[j + 1]; ++k) {
      const int i = M.i[k];
      const real a = r_abs(M.x[k]);
      E[j] = r_max(a, E[j]);
      if (i != j) E[i] = r_max(a, E[i]);
    }
}
static void premult_diag(CscR& M, const rvec& d) {
  for (int j = 0; j < M.n; ++j)
    for (int k = M.p[j]; k < M.p[j + 1]; ++k) M.x[k] *= d[M.i[k]];
}
static void postmult_diag(CscR& M, const rvec& d) {
  for (int j = 0; j < M.n; ++j)
    for (int k = M.p[j]; k < M.p[j + 1]; ++k) M.x[k] *= d[j];
}
static void limit_scaling(rvec& D) {
  for (real& d : D) {
    d = d < MIN_SCALING ? 1.0 : d;
    d = d > MAX_SCALING ? MAX_SCALING : d;
  }
}
static real limit_scaling1(real d) {
  d = d < MIN_SCALING ? 1.0 : d;
  d = d > MAX_SCALING ? MAX_SCALING : d;
  return d;
}

// ---------------------------------------------------------------------------------------------------------
// QDLDL restated: elimination tree, up-looking numeric LDL^T, triangular solves.
// ---------------------------------------------------------------------------------------------------------
struct Ldl {
  int n = 0;
  std::vector<int> etree, Lnz, Lp, Li;
  rvec Lx, D, Dinv;
  // work
  std::vector<int> ymark, yidx, ebuf, lnext;
  rvec yvals;

  bool symbolic(const CscR& K) {
    n = K.n;
    etree.assign(n, -1);
    Lnz.assign(n, 0);
    std::vector<int> work(n, 0);
    for (int j = 0; j < n; ++j) {
      work[j] = j;
      for (int p = K.p[j]; p < K.p[j + 1]; ++p) {
        int i = K.i[p];
        if (i > j) return false;  // not upper triangular
        while (work[i] != j) {
          if (etree[i] == -1) etree[i] = j;
          Lnz[i]++;
          work[i] = j;
          i = etree[i];
        }
      }
    }
    Lp.assign(n + 1, 0);
    for (int i = 0; i < n; ++i) Lp[i + 1] = Lp[i] + Lnz[i];
    Li.assign(Lp[n], 0);
    Lx.assign(Lp[n], 0.0);
    D.assign(n, 0.0);
    Dinv.assign(n, 0.0);
    ymark.assign(n, 0);
    yidx.assign(n, 0);
    ebuf.assign(n, 0);
    lnext.assign(n, 0);
    yvals.assign(n, 0.0);
    return true;
  }

  // returns number of positive pivots, or -1 on a zero pivot
  int numeric(const CscR& K) {
    int positive = 0;
    for (int i = 0; i < n; ++i) {
      ymark[i] = 0;
      yvals[i] = 0.0;
      D[i] = 0.0;
      lnext[i] = Lp[i];
    }
    for (int k = 0; k < n; ++k) {
      int nnzY = 0;
      for (int p = K.p[k]; p < K.p[k + 1]; ++p) {
        const int b = K.i[p];
        if (b == k) {
          D[k] = K.x[p];
          continue;
        }
        yvals[b] = K.x[p];
        int nxt = b;
        if (!ymark[nxt]) {
          ymark[nxt] = 1;
          int nE = 0;
          ebuf[nE++] = nxt;
          nxt = etree[b];
          while (nxt != -1 && nxt < k) {
            if (ymark[nxt]) break;
            ymark[nxt] = 1;
            ebuf[nE++] = nxt;
            nxt = etree[nxt];
          }
          while (nE) yidx[nnzY++] = ebuf[--nE];
        }
      }
      for (int q = nnzY - 1; q >= 0; --q) {
        const int c = yidx[q];
        const int tmp = lnext[c];
        const real yc = yvals[c];
        for (int j = Lp[c]; j < tmp; ++j) yvals[Li[j]] -= Lx[j] * yc;
        Li[tmp] = k;
        Lx[tmp] = yc * Dinv[c];
        D[k] -= yc * Lx[tmp];
        lnext[c]++;
        yvals[c] = 0.0;
        ymark[c] = 0;
      }
      if (D[k] == 0.0) return -1;
      if (D[k] > 0.0) positive++;
      Dinv[k] = 1.0 / D[k];
    }
    return positive;
  }

  void solve(rvec& x) const {
    for (int i = 0; i < n; ++i) {
      const real v = x[i];
      for (int j = Lp[i]; j < Lp[i + 1]; ++j) x[Li[j]] -= Lx[j] * v;
    }
    for (int i = 0; i < n; ++i) x[i] *= Dinv[i];
    for (int i = n - 1; i >= 0; --i) {
      real v = x[i];
      for (int j = Lp[i]; j < Lp[i + 1]; ++j) v -= Lx[j] * x[Li[j]];
      x[i] = v;
    }
  }
};

// KKT = [P + sigma I, A'; A, -diag(1/rho)] (kkt.c form_KKT, format 0), symmetrically permuted, upper triangle.
struct Kkt {
  int n = 0, m = 0;
  CscR K;                      // permuted upper triangle
  std::vector<int> perm;      // perm[k] = original index placed at position k
  std::vector<int> iperm;
  std::vector<int> rho_pos;   // position in K.x of the -1/rho_i diagonal entry
  Ldl ldl;
  rvec bp;

  void build(const CscR& P, const CscR& A, real sigma, const rvec& rho_inv,
             const std::vector<int>* var_order) {
    n = P.n;
    m = A.m;
    const int N = n + m;
    perm.resize(N);
    iperm.resize(N);
    // elimination order: constraint rows (degree <= 4) first, then variables in the caller's order
    for (int i = 0; i < m; ++i) perm[i] = n + i;
    for (int j = 0; j < n; ++j) perm[m + j] = var_order ? (*var_order)[j] : j;
    for (int k = 0; k < N; ++k) iperm[perm[k]] = k;

    TripletListR T;
    std::vector<char> has_diag(n, 0);
    auto put = [&](int r, int c, real v) {
      int a = iperm[r], b = iperm[c];
      if (a > b) std::swap(a, b);
      T.add(a, b, v);
    };
    for (int j = 0; j < n; ++j)
      for (int k = P.p[j]; k < P.p[j + 1]; ++k) {
        const int i = P.i[k];
        if (i == j) {
          has_diag[j] = 1;
          put(i, j, P.x[k] + sigma);
        } else {
          put(i, j, P.x[k]);
        }
      }
    for (int j = 0; j < n; ++j)
      if (!has_diag[j]) put(j, j, sigma);
    for (int j = 0; j < n; ++j)
      for (int k = A.p[j]; k < A.p[j + 1]; ++k) put(j, n + A.i[k], A.x[k]);
    for (int i = 0; i < m; ++i) put(n + i, n + i, -rho_inv[i]);
    K = csc_from_triplets_t<CscR, TripletListR>(N, N, T);
    rho_pos.assign(m, -1);
    for (int i = 0; i < m; ++i) {
      const int c = iperm[n + i];
      for (int k = K.p[c]; k < K.p[c + 1]; ++k)
        if (K.i[k] == c) rho_pos[i] = k;
      assert(rho_pos[i] >= 0);
    }
    bool ok = ldl.symbolic(K);
    assert(ok);
    (void)ok;
    bp.resize(N);
    g_last_factor_nnz = ldl.Lp[N];
  }
  bool factor() { return ldl.numeric(K) >= 0; }
  void update_rho(const rvec& rho_inv) {
    for (int i = 0; i < m; ++i) K.x[rho_pos[i]] = -rho_inv[i];
  }
  // LDLSolve of qdldl_interface.c: permute, solve, un-permute
  void solve(rvec& b) {
    const int N = n + m;
    for (int k = 0; k < N; ++k) bp[k] = b[perm[k]];
    ldl.solve(bp);
    for (int k = 0; k < N; ++k) b[perm[k]] = bp[k];
  }
};

// ---------------------------------------------------------------------------------------------------------
// Workspace mirroring OSQPWorkspace for one solve
// ---------------------------------------------------------------------------------------------------------
struct Work {
  int n, m;
  CscR P, A;  // scaled copies
  rvec q, l, u;
  rvec D, Dinv, E, Einv;
  real c = 1.0, cinv = 1.0;
  rvec rho_vec, rho_inv_vec;
  std::vector<int> constr_type;
  real rho;
  rvec x, y, z, xz_tilde, x_prev, z_prev, Ax, Px, Aty, delta_y, Atdelta_y, delta_x, Pdelta_x,
      Adelta_x;
  Kkt kkt;
  Info info;
  real pri_res = 0, dua_res = 0;  // info.pri_res / info.dua_res in the solve's arithmetic (the tests read these)
};

// scaling.c scale_data
static void scale_data(Work& w, int passes) {
  const int n = w.n, m = w.m;
  w.c = 1.0;
  w.D.assign(n, 1.0);
  w.Dinv.assign(n, 1.0);
  w.E.assign(m, 1.0);
  w.Einv.assign(m, 1.0);
  rvec Dt(n), DtA(n), Et(m);
  for (int it = 0; it < passes; ++it) {
    // norms of the KKT columns [P;A] and [A';0]
    inf_norm_cols_sym_triu(w.P, Dt);
    inf_norm_cols(w.A, DtA);
    for (int j = 0; j < n; ++j) Dt[j] = r_max(Dt[j], DtA[j]);
    inf_norm_rows(w.A, Et);
    limit_scaling(Dt);
    limit_scaling(Et);
    for (real& d : Dt) d = 1.0 / r_sqrt(d);
    for (real& e : Et) e = 1.0 / r_sqrt(e);
    premult_diag(w.P, Dt);
    postmult_diag(w.P, Dt);
    premult_diag(w.A, Et);
    postmult_diag(w.A, Dt);
    for (int j = 0; j < n; ++j) w.q[j] = Dt[j] * w.q[j];
    for (int j = 0; j < n; ++j) w.D[j] = w.D[j] * Dt[j];
    for (int i = 0; i < m; ++i) w.E[i] = w.E[i] * Et[i];
    // cost normalisation
    inf_norm_cols_sym_triu(w.P, Dt);
    real c_temp = 0.0;
    for (int j = 0; j < n; ++j) c_temp += Dt[j];
    c_temp /= (real)n;
    real inf_norm_q = limit_scaling1(vec_norm_inf(w.q));
    c_temp = r_max(c_temp, inf_norm_q);
    c_temp = limit_scaling1(c_temp);
    c_temp = 1.0 / c_temp;
    for (real& v : w.P.x) v *= c_temp;
    for (real& v : w.q) v *= c_temp;
    w.c *= c_temp;
  }
  w.cinv = 1.0 / w.c;
  for (int j = 0; j < n; ++j) w.Dinv[j] = 1.0 / w.D[j];
  for (int i = 0; i < m; ++i) w.Einv[i] = 1.0 / w.E[i];
  for (int i = 0; i < m; ++i) w.l[i] = w.E[i] * w.l[i];
  for (int i = 0; i < m; ++i) w.u[i] = w.E[i] * w.u[i];
}

// auxil.c set_rho_vec
static void set_rho_vec(Work& w) {
  w.rho = r_min(r_max(w.rho, RHO_MIN), RHO_MAX);
  for (int i = 0; i < w.m; ++i) {
    if (w.l[i] < -OSQP_INFTY * MIN_SCALING && w.u[i] > OSQP_INFTY * MIN_SCALING) {
      w.constr_type[i] = -1;
      w.rho_vec[i] = RHO_MIN;
    } else if (w.u[i] - w.l[i] < RHO_TOL) {
      w.constr_type[i] = 1;
      w.rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w.rho;
    } else {
      w.constr_type[i] = 0;
      w.rho_vec[i] = w.rho;
    }
    w.rho_inv_vec[i] = 1.0 / w.rho_vec[i];
  }
}

// auxil.c compute_pri_res / compute_dua_res (leave Ax-z in z_prev and Px+q+A'y in x_prev, as upstream does)
static real compute_pri_res(Work& w) {
  mat_vec(w.A, w.x, w.Ax, false);
  for (int i = 0; i < w.m; ++i) w.z_prev[i] = w.Ax[i] - w.z[i];
  return vec_scaled_norm_inf(w.Einv, w.z_prev);
}
static real compute_dua_res(Work& w) {
  w.x_prev = w.q;
  mat_vec(w.P, w.x, w.Px, false);
  mat_tpose_vec(w.P, w.x, w.Px, true, true);
  for (int j = 0; j < w.n; ++j) w.x_prev[j] = w.x_prev[j] + w.Px[j];
  if (w.m > 0) {
    mat_tpose_vec(w.A, w.y, w.Aty, false, false);
    for (int j = 0; j < w.n; ++j) w.x_prev[j] = w.x_prev[j] + w.Aty[j];
  }
  return w.cinv * vec_scaled_norm_inf(w.Dinv, w.x_prev);
}
static void update_info(Work& w, int iter) {
  w.info.iter = iter;
  w.pri_res = (w.m == 0) ? 0.0 : compute_pri_res(w);
  w.dua_res = compute_dua_res(w);
}
static real compute_pri_tol(const Work& w, real eps_abs, real eps_rel) {
  real mx = vec_scaled_norm_inf(w.Einv, w.z);
  mx = r_max(mx, vec_scaled_norm_inf(w.Einv, w.Ax));
  return eps_abs + eps_rel * mx;
}
static real compute_dua_tol(const Work& w, real eps_abs, real eps_rel) {
  real mx = vec_scaled_norm_inf(w.Dinv, w.q);
  mx = r_max(mx, vec_scaled_norm_inf(w.Dinv, w.Aty));
  mx = r_max(mx, vec_scaled_norm_inf(w.Dinv, w.Px));
  mx *= w.cinv;
  return eps_abs + eps_rel * mx;
}

// auxil.c is_primal_infeasible.  NOTE (reference quirk): the reference passes a true -infinity lower bound
// for inter-vehicle rows (sqp/dsqp_solver.cc:1121-1123); l_i * min(dy_i,0) is then (-inf)*0 = NaN and the
// certificate test below is false for every agent that has inter-vehicle rows.  IEEE semantics reproduce that.
static bool is_primal_infeasible(Work& w, real eps_prim_inf) {
  for (int i = 0; i < w.m; ++i) {
    if (w.u[i] > OSQP_INFTY * MIN_SCALING) {
      if (w.l[i] < -OSQP_INFTY * MIN_SCALING)
        w.delta_y[i] = 0.0;
      else
        w.delta_y[i] = r_min(w.delta_y[i], 0.0);
    } else if (w.l[i] < -OSQP_INFTY * MIN_SCALING) {
      w.delta_y[i] = r_max(w.delta_y[i], 0.0);
    }
  }
  for (int i = 0; i < w.m; ++i) w.Adelta_x[i] = w.E[i] * w.delta_y[i];
  const real norm_dy = vec_norm_inf(w.Adelta_x);
  if (norm_dy > eps_prim_inf) {
    real lhs = 0.0;
    for (int i = 0; i < w.m; ++i)
      lhs += w.u[i] * r_max(w.delta_y[i], 0.0) + w.l[i] * r_min(w.delta_y[i], 0.0);
    if (lhs < -eps_prim_inf * norm_dy) {
      mat_tpose_vec(w.A, w.delta_y, w.Atdelta_y, false, false);
      for (int j = 0; j < w.n; ++j) w.Atdelta_y[j] = w.Dinv[j] * w.Atdelta_y[j];
      return vec_norm_inf(w.Atdelta_y) < eps_prim_inf * norm_dy;
    }
  }
  return false;
}

// auxil.c is_dual_infeasible
static bool is_dual_infeasible(Work& w, real eps_dual_inf) {
  const real norm_dx = vec_scaled_norm_inf(w.D, w.delta_x);
  const real cost_scaling = w.c;
  if (norm_dx > eps_dual_inf) {
    real qdx = 0.0;
    for (int j = 0; j < w.n; ++j) qdx += w.q[j] * w.delta_x[j];
    if (qdx < -cost_scaling * eps_dual_inf * norm_dx) {
      mat_vec(w.P, w.delta_x, w.Pdelta_x, false);
      mat_tpose_vec(w.P, w.delta_x, w.Pdelta_x, true, true);
      for (int j = 0; j < w.n; ++j) w.Pdelta_x[j] = w.Dinv[j] * w.Pdelta_x[j];
      if (vec_norm_inf(w.Pdelta_x) < cost_scaling * eps_dual_inf * norm_dx) {
        mat_vec(w.A, w.delta_x, w.Adelta_x, false);
        for (int i = 0; i < w.m; ++i) w.Adelta_x[i] = w.Einv[i] * w.Adelta_x[i];
        for (int i = 0; i < w.m; ++i) {
          if ((w.u[i] < OSQP_INFTY * MIN_SCALING && w.Adelta_x[i] > eps_dual_inf * norm_dx) ||
              (w.l[i] > -OSQP_INFTY * MIN_SCALING && w.Adelta_x[i] < -eps_dual_inf * norm_dx))
            return false;
        }
        return true;
      }
    }
  }
  return false;
}

// auxil.c check_termination
static bool check_termination(Work& w, const Settings& st, bool approximate) {
  real eps_abs = st.eps_abs, eps_rel = st.eps_rel, eps_pinf = st.eps_prim_inf, eps_dinf = st.eps_dual_inf;
  bool prim_res_check = false, dual_res_check = false, prim_inf_check = false, dual_inf_check = false;
  if (w.pri_res > OSQP_INFTY || w.dua_res > OSQP_INFTY) {
    w.info.status = NON_CVX;
    return true;
  }
  if (approximate) {
    eps_abs *= 10;
    eps_rel *= 10;
    eps_pinf *= 10;
    eps_dinf *= 10;
  }
  if (w.m == 0) {
    prim_res_check = true;
  } else {
    const real eps_prim = compute_pri_tol(w, eps_abs, eps_rel);
    if (w.pri_res < eps_prim)
      prim_res_check = true;
    else
      prim_inf_check = is_primal_infeasible(w, eps_pinf);
  }
  const real eps_dual = compute_dua_tol(w, eps_abs, eps_rel);
  if (w.dua_res < eps_dual)
    dual_res_check = true;
  else
    dual_inf_check = is_dual_infeasible(w, eps_dinf);

  if (prim_res_check && dual_res_check) {
    w.info.status = approximate ? SOLVED_INACCURATE : SOLVED;
    return true;
  } else if (prim_inf_check) {
    w.info.status = approximate ? PRIMAL_INFEASIBLE_INACCURATE : PRIMAL_INFEASIBLE;
    return true;
  } else if (dual_inf_check) {
    w.info.status = approximate ? DUAL_INFEASIBLE_INACCURATE : DUAL_INFEASIBLE;
    return true;
  }
  return false;
}

// auxil.c compute_rho_estimate (uses the scaled residual vectors left in z_prev / x_prev by update_info)
static real compute_rho_estimate(const Work& w, const Settings&) {
  real pri_res = vec_norm_inf(w.z_prev);
  real dua_res = vec_norm_inf(w.x_prev);
  real pri_norm = r_max(vec_norm_inf(w.z), vec_norm_inf(w.Ax));
  pri_res /= (pri_norm + 1e-10);
  real dua_norm = vec_norm_inf(w.q);
  dua_norm = r_max(dua_norm, vec_norm_inf(w.Aty));
  dua_norm = r_max(dua_norm, vec_norm_inf(w.Px));
  dua_res /= (dua_norm + 1e-10);
  real est = w.rho * r_sqrt(pri_res / (dua_res + 1e-10));
  est = r_min(r_max(est, RHO_MIN), RHO_MAX);
  return est;
}

Info osqp_solve_restated(const Csc& P_triu, const std::vector<double>& q, const Csc& A,
                         const std::vector<double>& l, const std::vector<double>& u,
                         const std::vector<double>& x_warm, const Settings& st, std::vector<double>& x_out,
                         std::vector<double>* y_out, Trace* trace, const std::vector<int>* var_order) {
  Work w;
  const int n = w.n = P_triu.n;
  const int m = w.m = A.m;
  // ---- osqp_setup ----
  w.P = to_real(P_triu);
  w.A = to_real(A);
  w.q = to_real(q);
  w.l = to_real(l);
  w.u = to_real(u);
  w.rho = st.rho;
  w.rho_vec.assign(m, 0.0);
  w.rho_inv_vec.assign(m, 0.0);
  w.constr_type.assign(m, 0);
  w.x.assign(n, 0.0);
  w.y.assign(m, 0.0);
  w.z.assign(m, 0.0);
  w.xz_tilde.assign(n + m, 0.0);
  w.x_prev.assign(n, 0.0);
  w.z_prev.assign(m, 0.0);
  w.Ax.assign(m, 0.0);
  w.Px.assign(n, 0.0);
  w.Aty.assign(n, 0.0);
  w.delta_y.assign(m, 0.0);
  w.Atdelta_y.assign(n, 0.0);
  w.delta_x.assign(n, 0.0);
  w.Pdelta_x.assign(n, 0.0);
  w.Adelta_x.assign(m, 0.0);
  if (st.scaling) {
    scale_data(w, st.scaling);
  } else {
    w.D.assign(n, 1.0);
    w.Dinv.assign(n, 1.0);
    w.E.assign(m, 1.0);
    w.Einv.assign(m, 1.0);
  }
  set_rho_vec(w);
  w.kkt.build(w.P, w.A, st.sigma, w.rho_inv_vec, var_order);
  bool fac_ok = w.kkt.factor();
  assert(fac_ok);
  (void)fac_ok;
  w.info.status = UNSOLVED;

  // ---- osqp_warm_start_x: x <- Dinv x0, z <- A x, y stays 0 ----
  for (int j = 0; j < n; ++j) w.x[j] = w.Dinv[j] * real(x_warm[j]);
  mat_vec(w.A, w.x, w.z, false);

  // ---- osqp_solve ----
  int iter;
  bool can_check = false;
  for (iter = 1; iter <= st.max_iter; ++iter) {
    std::swap(w.x, w.x_prev);
    std::swap(w.z, w.z_prev);
    // update_xz_tilde: rhs then KKT solve (qdldl_interface.c solve_linsys_qdldl)
    for (int j = 0; j < n; ++j) w.xz_tilde[j] = st.sigma * w.x_prev[j] - w.q[j];
    for (int i = 0; i < m; ++i) w.xz_tilde[n + i] = w.z_prev[i] - w.rho_inv_vec[i] * w.y[i];
    {
      rvec sol = w.xz_tilde;
      w.kkt.solve(sol);
      for (int j = 0; j < n; ++j) w.xz_tilde[j] = sol[j];
      for (int i = 0; i < m; ++i) w.xz_tilde[n + i] += w.rho_inv_vec[i] * sol[n + i];
    }
    // update_x
    for (int j = 0; j < n; ++j) w.x[j] = st.alpha * w.xz_tilde[j] + (1.0 - st.alpha) * w.x_prev[j];
    for (int j = 0; j < n; ++j) w.delta_x[j] = w.x[j] - w.x_prev[j];
    // update_z + project
    for (int i = 0; i < m; ++i)
      w.z[i] = st.alpha * w.xz_tilde[n + i] + (1.0 - st.alpha) * w.z_prev[i] + w.rho_inv_vec[i] * w.y[i];
    for (int i = 0; i < m; ++i) w.z[i] = r_min(r_max(w.z[i], w.l[i]), w.u[i]);
    // update_y
    for (int i = 0; i < m; ++i) {
      w.delta_y[i] = w.rho_vec[i] * (st.alpha * w.xz_tilde[n + i] + (1.0 - st.alpha) * w.z_prev[i] - w.z[i]);
      w.y[i] += w.delta_y[i];
    }

    can_check = st.check_termination && (iter % st.check_termination == 0);
    if (can_check) {
      update_info(w, iter);
      if (trace) {
        trace->pri_res_hist.push_back((double)w.pri_res);
        trace->dua_res_hist.push_back((double)w.dua_res);
        trace->rho_hist.push_back((double)w.rho);
      }
      if (check_termination(w, st, false)) break;
    }
    if (st.adaptive_rho && st.adaptive_rho_interval && (iter % st.adaptive_rho_interval == 0)) {
      if (!can_check) update_info(w, iter);
      const real rho_new = compute_rho_estimate(w, st);
      if (rho_new > w.rho * st.adaptive_rho_tolerance || rho_new < w.rho / st.adaptive_rho_tolerance) {
        // osqp_update_rho
        w.rho = r_min(r_max(rho_new, RHO_MIN), RHO_MAX);
        for (int i = 0; i < m; ++i) {
          if (w.constr_type[i] == 0) {
            w.rho_vec[i] = w.rho;
            w.rho_inv_vec[i] = 1.0 / w.rho;
          } else if (w.constr_type[i] == 1) {
            w.rho_vec[i] = RHO_EQ_OVER_RHO_INEQ * w.rho;
            w.rho_inv_vec[i] = 1.0 / w.rho_vec[i];
          }
        }
        w.kkt.update_rho(w.rho_inv_vec);
        bool ok = w.kkt.factor();
        assert(ok);
        (void)ok;
        w.info.rho_updates++;
      }
    }
  }
  if (!can_check) {
    update_info(w, iter - 1);
    check_termination(w, st, false);
  }
  if (w.info.status == UNSOLVED) {
    if (!check_termination(w, st, true)) w.info.status = MAX_ITER_REACHED;
  }
  if (iter > st.max_iter) iter = st.max_iter;
  w.info.iter = iter;
  w.info.rho_final = (double)w.rho;
  w.info.pri_res = (double)w.pri_res;
  w.info.dua_res = (double)w.dua_res;

  // store_solution / unscale_solution
  x_out.assign(n, std::numeric_limits<double>::quiet_NaN());
  if (y_out) y_out->assign(m, std::numeric_limits<double>::quiet_NaN());
  if (has_solution(w.info.status)) {
    for (int j = 0; j < n; ++j) x_out[j] = (double)(w.D[j] * w.x[j]);
    if (y_out)
      for (int i = 0; i < m; ++i) (*y_out)[i] = (double)(w.cinv * (w.E[i] * w.y[i]));
  }
  if (trace) {
    trace->D.assign(w.D.begin(), w.D.end());
    trace->E.assign(w.E.begin(), w.E.end());
    trace->c = (double)w.c;
    trace->l_s.assign(w.l.begin(), w.l.end());
    trace->u_s.assign(w.u.begin(), w.u.end());
    trace->rho_vec.assign(w.rho_vec.begin(), w.rho_vec.end());
    trace->x_scaled.assign(w.x.begin(), w.x.end());
    trace->y_scaled.assign(w.y.begin(), w.y.end());
    trace->z_scaled.assign(w.z.begin(), w.z.end());
  }
  return w.info;
}

}  // namespace csdo_oracle
