// dsqp_program.h — the per-agent DO-phase program: one workgroup per agent, one lane per timestep.
//
// What it computes (reference: sqp/dsqp_solver.cc of YangSVM/CSDOTrajectoryPlanning):
//   SolverDSQP::calcIndividualSQP (:36-269) for ONE agent, start to finish, without leaving the device:
//     initial safe corridors (sqp/corridor.cc:164-248) -> repeat { linearise + assemble the QP (:646-1129),
//     OSQP-0.6.3-style ADMM solve (the reference calls osqp_setup/osqp_solve at :457-502), delta test (:228),
//     feasibility test (:292-420), corridor refresh (:818-872) }.
//
// How it is laid out on CDNA4 (this is not how the reference or OSQP do it):
//   * lane t owns everything that belongs to timestep t: the 6 variables (x,y,yaw,steer,v,w)_t, the 16 constraint
//     rows whose "home" is t (4 kinematic rows t->t+1, 3 start/goal rows, 4 corridor rows, 2 trust rows, 2 control
//     rows, 1 steer row) with their ADMM state (y, z, bounds, scaling) IN REGISTERS, plus a loop over the
//     inter-vehicle rows at t whose state lives in an L2-resident workspace;
//   * the KKT system is never formed.  Eliminating the constraint block gives H = P + sigma I + A' diag(rho) A,
//     which in time-major order is block tridiagonal with 6x6 blocks.  It is factorised and solved by BLOCK
//     CYCLIC REDUCTION (log2 Nt levels, every level parallel over timesteps): lane t keeps the inverse of its
//     pivot block in registers, the coupling blocks sit in LDS (or in HBM/L2 when Nt is too long for 160 KB);
//   * neighbour data (t-1, t+1) and BCR partial vectors cross lanes through LDS; infinity norms for the
//     termination test are wave-shuffle + LDS reductions.
//
// The same source is compiled twice: as HIP device code (CSDO_LANE_MODE_DEVICE, csrc/dsqp_kernel.hip) and as a
// lane-serial host build (CSDO_LANE_MODE_SERIAL, tests/emu/) that runs the lanes of one agent in a loop so the CPU
// test-suite can check the program logic against the oracle without a GPU.  The serial build is test
// infrastructure; the shipped library contains only the device build and fails loudly without a GPU.
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>

#include "csdo_device_types.h"

#if defined(CSDO_LANE_MODE_DEVICE)
#include <hip/hip_runtime.h>
#define CSDO_FN __device__ __forceinline__
#define CSDO_LANES(t) if (const int t = (int)threadIdx.x; t < Nt)
#define CSDO_LS(t) lanes
#define CSDO_SYNC() __syncthreads()
#define CSDO_LANESTORE LaneState&
#define CSDO_MARK(name) asm volatile("; CSDO_MARK " name)
#elif defined(CSDO_LANE_MODE_SERIAL)
#define CSDO_FN inline
#define CSDO_LANES(t) for (int t = 0; t < Nt; ++t)
#define CSDO_LS(t) lanes[t]
#define CSDO_SYNC() ((void)0)
#define CSDO_LANESTORE LaneState*
#define CSDO_MARK(name) ((void)0)
#else
#error "define CSDO_LANE_MODE_DEVICE or CSDO_LANE_MODE_SERIAL"
#endif

namespace csdo {

// ---------------------------------------------------------------------------------------------------------
// Row / column tables.  Home rows of timestep t (SURVEY Appendix A for the coefficients):
//   0 x-dyn  1 y-dyn  2 yaw-dyn  3 steer-dyn   (t <= Nt-2; each also touches ONE column of t+1, coefficient cn)
//   4 cfg-x  5 cfg-y  6 cfg-yaw               (t == 0 or t == Nt-1)
//   7 xf  8 yf  9 xr  10 yr                   corridor
//   11 trust-x  12 trust-y   13 |v|  14 |w| (t <= Nt-2)   15 |steer|
// Columns of timestep t: 0 x, 1 y, 2 yaw, 3 steer, 4 v, 5 w (v, w exist for t <= Nt-2).
// ---------------------------------------------------------------------------------------------------------
constexpr int NROW = 16;
constexpr int NCOLS = 6;

CSDO_FN constexpr int row_col(int i, int s) {
  constexpr int T[NROW][3] = {{0, 2, 4}, {1, 2, 4}, {2, 3, 4}, {3, 5, -1}, {0, -1, -1}, {1, -1, -1},
                              {2, -1, -1}, {0, 2, -1}, {1, 2, -1}, {0, 2, -1}, {1, 2, -1}, {0, -1, -1},
                              {1, -1, -1}, {4, -1, -1}, {5, -1, -1}, {3, -1, -1}};
  return T[i][s];
}
// kin row i (0..3) touches column i of timestep t+1
constexpr unsigned ROWS_KIN = 0x000Fu, ROWS_CFG = 0x0070u, ROWS_CTRL = 0x6000u;
constexpr unsigned ROWS_ALWAYS_EQ = 0x007Fu;

constexpr double OSQP_INFTY = 1e30, RHO_MIN = 1e-6, RHO_MAX = 1e6, RHO_TOL = 1e-4, RHO_EQ_OVER_RHO_INEQ = 1e3;
constexpr double MIN_SCALING = 1e-4, MAX_SCALING = 1e4;

template <int N, class F>
CSDO_FN void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}
#define CSDO_FOR(I, N, ...) static_for<N>([&](auto I##_c) __attribute__((always_inline)) { constexpr int I = decltype(I##_c)::value; __VA_ARGS__ })

CSDO_FN double dmax(double a, double b) { return (b > a) ? b : a; }   // NaN in b is ignored, like vec_norm_inf
CSDO_FN double dmin(double a, double b) { return (b < a) ? b : a; }
CSDO_FN double osqp_max(double a, double b) { return (a > b) ? a : b; }  // c_max
CSDO_FN double osqp_min(double a, double b) { return (a < b) ? a : b; }  // c_min
CSDO_FN double limit_scaling(double d) {
  d = d < MIN_SCALING ? 1.0 : d;
  d = d > MAX_SCALING ? MAX_SCALING : d;
  return d;
}
CSDO_FN constexpr int sym(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }

// ---------------------------------------------------------------------------------------------------------
// Per-lane state (registers on the device)
// ---------------------------------------------------------------------------------------------------------
struct LaneState {          // ADMM-hot state: stays in registers for the whole QP
  double c[NROW][3];        // scaled coefficients on own columns
  double cn[4];             // scaled coefficient of kin rows on column i of t+1
  double lo[NROW], hi[NROW];// scaled bounds
  double y[NROW], z[NROW];  // ADMM dual / slack (z doubles as Ruiz scratch before the warm start)
  double x[6];              // scaled primal iterate
  double b[6];              // rhs -> BCR work vector -> x_tilde
  double Pvv, Pww, Pvn;     // scaled objective: diag v, diag w, coupling (v_t, v_{t+1})
  double sinv[21];          // inverse of this node's BCR pivot block (symmetric, packed lower)
  double fa[21], fr[36];    // factor-time only: current diagonal block and coupling to the right neighbour
  unsigned eqmask;          // rows in OSQP's "equality" class (rho * 1e3)
  unsigned loosemask;       // rows with both bounds infinite (rho = RHO_MIN)
  unsigned act;             // rows that exist at this t
  int ncols;                // 6, or 4 at t = Nt-1
};

// Cold per-lane data lives in a coalesced per-agent workspace [slot][stride] (HBM, L2-resident): touched once per
// SQP iteration or once per termination check, never inside the ADMM iteration.
enum ColdSlot {
  C_E = 0,      // [16] Ruiz row scaling
  C_D = 16,     // [6]  Ruiz column scaling
  C_SOL0 = 22,  // [6]  previous SQP iterate == linearisation point (x,y,yaw,steer,v,w), unscaled
  C_SOL = 28,   // [6]  latest QP solution, unscaled
  C_XT = 34, C_YT = 35, C_YAWT = 36,  // original initial guess (trust centre, start/goal rows): never moved
  C_CLB = 37,   // [4]  corridor of the linearisation point: xf, yf, xr, yr lower
  C_CUB = 41,   // [4]  upper
  C_DY = 45,    // [16] last y-increment, kept on termination-check iterations only
  C_TOTAL = 61
};

// Shared (LDS) arrays, SoA with `stride` doubles per component
struct Shm {
  double* vec;      // [6][stride]  x_tilde / x exchange
  double* pl;       // [6][stride]  BCR partials for the left neighbour; aliases red[0..5]
  double* pr;       // [6][stride]  BCR partials for the right neighbour; aliases red[6..11]
  double* carry;    // [6][stride]  t -> t+1 hand-over (rhs, norms, scalings)
  double* carry2;   // [6][stride]  t -> t-1 hand-over
  double* obs;      // [3][n_obs]
  double* facE;     // [72][stride] coupling blocks (LDS or global)
  double* facX;     // [78][stride] factor-time exchange (global)
  double* cold;     // [C_TOTAL][stride] cold per-lane data (global)
  double* bcast;    // [32] block-wide results
  int stride;
};

struct AgentCtx {
  int Nt, Nm;
  const double* x0;
  const PlaneDev* planes;
  const int32_t* tstart;
  double* rows;  // inter-row workspace of this agent: [4K][8]
  double dimx, dimy;
  int n_obs;
};

// inter-row workspace accessors (AoS of 8 doubles per row)
enum { W_Y = 0, W_Z = 1, W_U = 2, W_E = 3, W_CA = 4, W_CB = 5, W_CY = 6, W_DY = 7 };

// =========================================================================================================
// Block-wide reductions.  Partials are stored per lane in sh.pl/sh.pr (12 slots); after a barrier the first wave
// folds them.  max is order independent; sums use a fixed order (stride-64 serial, then a halving tree) that the
// serial build reproduces exactly, so device and emulation agree bit for bit.
// =========================================================================================================
template <int K>
CSDO_FN void red_put(const Shm& sh, int t, const double (&part)[K]) {
  static_assert(K <= 12, "reduction scratch holds 12 values per lane");
  CSDO_FOR(k, K, { sh.pl[k * sh.stride + t] = part[k]; });
}

#if defined(CSDO_LANE_MODE_DEVICE)
CSDO_FN double wave_shfl_down(double v, int off) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_down(lo, off, 64);
  hi = __shfl_down(hi, off, 64);
  return __hiloint2double(hi, lo);
}
#endif

// out[k] = reduction over lanes 0..Nt-1 of slot k.  IS_SUM=false: max with NaNs ignored.  Collective.
template <int K, bool IS_SUM>
CSDO_FN void red_fold(const Shm& sh, int Nt, double (&out)[K]) {
  CSDO_SYNC();
#if defined(CSDO_LANE_MODE_DEVICE)
  const int tid = (int)threadIdx.x;
  if (tid < 64) {
    CSDO_FOR(k, K, {
      double acc = 0.0;
      for (int j = tid; j < Nt; j += 64) {
        const double v = sh.pl[k * sh.stride + j];
        acc = IS_SUM ? (acc + v) : dmax(acc, v);
      }
      for (int off = 32; off >= 1; off >>= 1) {
        const double o = wave_shfl_down(acc, off);
        acc = IS_SUM ? (acc + o) : dmax(acc, o);
      }
      if (tid == 0) sh.bcast[k] = acc;
    });
  }
  __syncthreads();
  CSDO_FOR(k, K, { out[k] = sh.bcast[k]; });
  __syncthreads();
#else
  CSDO_FOR(k, K, {
    double lane[64];
    for (int l = 0; l < 64; ++l) {
      double acc = 0.0;
      for (int j = l; j < Nt; j += 64) {
        const double v = sh.pl[k * sh.stride + j];
        acc = IS_SUM ? (acc + v) : dmax(acc, v);
      }
      lane[l] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1)
      for (int l = 0; l < off; ++l) lane[l] = IS_SUM ? (lane[l] + lane[l + off]) : dmax(lane[l], lane[l + off]);
    out[k] = lane[0];
  });
#endif
}

// =========================================================================================================
// Safe-corridor boxes (sqp/corridor.cc of the reference), one lane per point.
// =========================================================================================================
struct BoxD {
  double x_min, y_min, x_max, y_max;
};

CSDO_FN bool box_valid(const BoxD& b, const double* obs, int n_obs, double dimx, double dimy, double rv) {
  // isBoxValid, corridor.cc:252-272: inside [rv, dim-rv]^2 and no obstacle centre strictly inside the box
  // inflated by (r_obs + rv) on every side
  if (b.x_min < rv || b.x_max > dimx - rv || b.y_min < rv || b.y_max > dimy - rv) return false;
  for (int k = 0; k < n_obs; ++k) {
    const double ox = obs[k], oy = obs[n_obs + k], infl = obs[2 * n_obs + k] + rv;
    if ((b.x_min - infl) < ox && ox < (b.x_max + infl) && (b.y_min - infl) < oy && oy < (b.y_max + infl))
      return false;
  }
  return true;
}

// generateLocalBox, corridor.cc:278-324: grow by 0.1 in the order +y, -x, -y, +x until blocked or >= 10 m
CSDO_FN bool grow_box(double xc, double yc, const double* obs, int n_obs, double dimx, double dimy, double rv,
                      BoxD& res) {
  const double ds = 0.1, l_limit = 10.0;
  BoxD box{xc, yc, xc, yc};
  double len0 = 0, len1 = 0, len2 = 0, len3 = 0;
  bool on0 = true, on1 = true, on2 = true, on3 = true;
  int num_expand = 0;
  while (on0 || on1 || on2 || on3) {
    if (on0) {
      BoxD tr = box;
      tr.y_max += ds;
      if (box_valid(tr, obs, n_obs, dimx, dimy, rv)) {
        num_expand++;
        len0 += ds;
        box = tr;
        if (len0 >= l_limit) on0 = false;
      } else {
        on0 = false;
      }
    }
    if (on1) {
      BoxD tr = box;
      tr.x_min -= ds;
      if (box_valid(tr, obs, n_obs, dimx, dimy, rv)) {
        num_expand++;
        len1 += ds;
        box = tr;
        if (len1 >= l_limit) on1 = false;
      } else {
        on1 = false;
      }
    }
    if (on2) {
      BoxD tr = box;
      tr.y_min -= ds;
      if (box_valid(tr, obs, n_obs, dimx, dimy, rv)) {
        num_expand++;
        len2 += ds;
        box = tr;
        if (len2 >= l_limit) on2 = false;
      } else {
        on2 = false;
      }
    }
    if (on3) {
      BoxD tr = box;
      tr.x_max += ds;
      if (box_valid(tr, obs, n_obs, dimx, dimy, rv)) {
        num_expand++;
        len3 += ds;
        box = tr;
        if (len3 >= l_limit) on3 = false;
      } else {
        on3 = false;
      }
    }
  }
  res = box;
  return num_expand > 0;
}

// generateBox, corridor.cc:124-159.  Returns bit0 = success, bits1-2 = initial status (0 legal, 1 out of map,
// 2 inside an inflated obstacle).  First colliding obstacle = lowest input index (documented deviation from the
// reference's unordered_set iteration order).
CSDO_FN int make_box(double x, double y, const double* obs, int n_obs, double dimx, double dimy, double rv,
                     BoxD& res) {
  int initial = 0;
  if (x < rv || x > dimx - rv || y < rv || y > dimy - rv) {  // isPointOutOfMap + projectNearBorder
    initial = 1;
    const double eps = 1e-3, x0 = x, y0 = y;
    if (x0 < rv) x = rv + eps;
    else if (x0 > dimx - rv) x = dimx - rv - eps;
    if (y0 < rv) y = rv + eps;
    else if (y0 > dimy - rv) y = dimy - rv - eps;
  }
  int hit = -1;
  for (int k = 0; k < n_obs; ++k) {  // isPointCollision
    const double ox = obs[k], oy = obs[n_obs + k], infl = obs[2 * n_obs + k] + rv;
    if ((x - infl) < ox && ox < (x + infl) && (y - infl) < oy && oy < (y + infl)) {
      hit = k;
      break;
    }
  }
  bool success;
  if (hit >= 0) {  // generateLegalPoint, corridor.cc:84-122
    initial = 2;
    const double hx = obs[hit], hy = obs[n_obs + hit], hr = obs[2 * n_obs + hit];
    const double theta0 = atan2(y - hy, x - hx);
    const double d = rv + hr + 0.2;
    success = false;
    BoxD cand{x, y, x, y};
    for (int i = 0; i < 20 && !success; ++i) {
      int j = i / 2;
      if (i % 2 == 1) j = -j;
      const double theta = theta0 + j * 2 * M_PI / 20;
      x = hx + d * cos(theta);
      y = hy + d * sin(theta);
      if (x > rv && x < dimx - rv && y > rv && y < dimy - rv) {
        grow_box(x, y, obs, n_obs, dimx, dimy, rv, cand);
        if (box_valid(cand, obs, n_obs, dimx, dimy, rv)) success = true;
      }
    }
    if (!success) cand = BoxD{x, y, x, y};  // zero-area fallback
    res = cand;
  } else {
    success = grow_box(x, y, obs, n_obs, dimx, dimy, rv, res);
  }
  return (success ? 1 : 0) | (initial << 1);
}

// =========================================================================================================
// 6x6 dense helpers (all indices compile-time so everything stays in registers)
// =========================================================================================================
// inverse of a symmetric positive definite 6x6 given as packed lower A[21]; LDL^T without pivoting
CSDO_FN void spd_inverse6(const double (&A)[21], double (&inv)[21]) {
  double L[6][6];
  double d[6], dinv[6];
  CSDO_FOR(j, 6, {
    double dj = A[sym(j, j)];
    CSDO_FOR(k, j, { dj = fma(-L[j][k] * d[k], L[j][k], dj); });
    d[j] = dj;
    dinv[j] = 1.0 / dj;
    CSDO_FOR(ii, 5 - j, {
      constexpr int i = j + 1 + ii;
      double v = A[sym(i, j)];
      CSDO_FOR(k, j, { v = fma(-L[i][k] * d[k], L[j][k], v); });
      L[i][j] = v * dinv[j];
    });
  });
  // M = L^{-1} (unit lower)
  double M[6][6];
  CSDO_FOR(j, 6, {
    CSDO_FOR(ii, 5 - j, {
      constexpr int i = j + 1 + ii;
      double v = -L[i][j];
      CSDO_FOR(kk, i - j - 1, {
        constexpr int k = j + 1 + kk;
        v = fma(-L[i][k], M[k][j], v);
      });
      M[i][j] = v;
    });
  });
  // inv = M' diag(dinv) M
  CSDO_FOR(r, 6, {
    CSDO_FOR(c, r + 1, {
      double v = (r == c) ? dinv[r] : M[r][c] * dinv[r];  // k = r term: M[r][r] = 1
      CSDO_FOR(kk, 5 - r, {
        constexpr int k = r + 1 + kk;
        v = fma(M[k][r] * dinv[k], M[k][c], v);
      });
      inv[sym(r, c)] = v;
    });
  });
}

CSDO_FN void symv6(const double (&S)[21], const double (&v)[6], double (&out)[6]) {
  CSDO_FOR(r, 6, {
    double a = 0.0;
    CSDO_FOR(c, 6, { a = fma(S[sym(r, c)], v[c], a); });
    out[r] = a;
  });
}

// =========================================================================================================
// The program
// =========================================================================================================
struct ProgramOut {
  int sqp_iters, admm_iters, last_status, static_legal;
};

// ---- A x for the home rows of lane t; xn = columns 0..3 of t+1 --------------------------------------------
CSDO_FN void rows_times_x(const LaneState& S, const double (&x)[6], const double (&xn)[4], double (&Ax)[NROW]) {
  CSDO_FOR(i, NROW, {
    double a = 0.0;
    CSDO_FOR(s, 3, {
      if constexpr (row_col(i, s) >= 0) a = fma(S.c[i][s], x[row_col(i, s)], a);
    });
    if constexpr (i < 4) a = fma(S.cn[i], xn[i], a);
    Ax[i] = a;
  });
}

CSDO_FN double rho_of(const LaneState& S, int i, double rho) {
  const unsigned bit = 1u << i;
  return (S.loosemask & bit) ? RHO_MIN : ((S.eqmask & bit) ? RHO_EQ_OVER_RHO_INEQ * rho : rho);
}

template <class LaneStore>
CSDO_FN void agent_program(const DeviceBatch& B, const int agent, const Shm& sh, LaneStore&& lanes,
                           ProgramOut& out);

}  // namespace csdo

#include "dsqp_program_impl.h"
