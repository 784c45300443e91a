// dsqp_program_impl.h — body of csdo::agent_program (see dsqp_program.h for the design).
// Included only from dsqp_program.h.
#pragma once

namespace csdo {

// LDS arrays are lane-major: element k of lane t at arr[t * LD + k] (one address register + immediate offsets; the
// strides 6 and 73 doubles are bank-conflict free for the b64/b128 reads of neighbouring and of 2h-strided lanes)
#define SH(arr, k, t) sh.arr[(t) * LD_##arr + (k)]
// factor-time exchange lives in global memory, coalesced [k][stride]
// (uniform base + field offset)[lane]: the field part stays scalar and the lane part is one unsigned 32-bit register, so
// the access is `global_load v, v_lane, s[base]` - no per-field 64-bit address in vector registers (the compiler
// otherwise hoists those out of the loops and spills them: a scratch reload in front of every workspace access)
#define SX(k, t) (sh.facX + (size_t)(k) * (size_t)csdo_opaque_s(sh.stride))[(unsigned)(t)]
#define FE(k, t) (sh.facE + (k))[(unsigned)(t) * (unsigned)FAC_E_DOUBLES]   // lane-major: one lane register + immediate offsets (SoA and tiles measured slower)
// F_r of node t: the solver lane's registers for the first ER_REG entries, LDS for the rest and for the pivot inverse;
// everything from the workspace for long horizons
// (MODE 2: the lean modes' layout - nothing of the factor in registers - with all of F_r in LDS, 36 doubles per timestep)
#define ER(k, t) FE(36 + (k), t)   /* one-lane form (mode 3): F_r of node t from the workspace */
#define SINV(k, t) ((MODE >= 1) ? WS(W_SINV + (k), t) : SH(fx, FX_ER + (k), t))
// set-up stage scratch, field-major over the ADMM block's exchange arrays (see "assemble the QP")
#define SU(k, t) sh.vec[(size_t)(k) * (size_t)csdo_opaque_s(sh.stride) + (unsigned)(t)]
// (the stride passes through an empty asm at every use: the per-slot base addresses - some sixty 64-bit scalars - are then formed
// where they are used, two scalar instructions each, instead of being hoisted out of the SQP loop and held, or spilled, across
// the ADMM iterations)
#define CD(slot, t) (sh.cold + (size_t)(slot) * (size_t)csdo_opaque_s(sh.stride))[(unsigned)(t)]
#define WS(slot, t) (sh.cold + (size_t)(slot) * (size_t)csdo_opaque_s(sh.stride))[(unsigned)(t)]
// mode 0, between ADMM blocks: y (0..15), z (16..31), x (32..37) of the iterate a block has just finished, field-major in the
// idle LDS arrays behind the six doubles per lane that the residual update's exchange (carry2) uses
#define HX(f, t) (sh.lohi + (size_t)(6 + (f)) * (size_t)csdo_opaque_s(sh.stride))[(unsigned)(t)]
// the iterate as the residual update right behind a block reads it
#define ITER_Y(i, t) ((MODE == 0) ? HX(i, t) : WS(W_Yv + (i), t))
#define ITER_Z(i, t) ((MODE == 0) ? HX(NROW + (i), t) : WS(W_Zv + (i), t))
#define ITER_X(j, t) ((MODE == 0) ? HX(2 * NROW + (j), t) : WS(W_X + (j), t))

// ---------------------------------------------------------------------------------------------------------
// Assembly of the home rows of timestep t at the linearisation point S.sol0 (unscaled values).
// Reference: calcKineConstraint :646-744, calcCfgConstraint :746-788, calcCorridorConstraint :874-968,
// calcTrustRegionConstraint :970-994, calcMaxCtrlAndSteerConstraint :996-1039, objective :163-197.
// ---------------------------------------------------------------------------------------------------------
CSDO_FN void assemble_home_rows(LaneState& S, const Shm& sh, int t, int Nt, const SolverParams& P, double& dyaw_f_x,
                                double& dyaw_f_y, double& dyaw_r_x, double& dyaw_r_y, double& e_xf, double& e_yf,
                                double& e_xr, double& e_yr) {
  const int Nm = Nt - 1;
  const double dt = P.dt, WB = P.WB;
  const double yaw = CD(C_SOL0 + 2, t), st = CD(C_SOL0 + 3, t), v = CD(C_SOL0 + 4, t);
  const double xt = CD(C_XT, t), yt = CD(C_YT, t), yawt = CD(C_YAWT, t);
  const SinCos scy_ = sincos_of(yaw);
  const double sy = scy_.s, cy = scy_.c;
  CSDO_FOR(i, NROW, {
    CSDO_FOR(s, 3, { S.c[i][s] = 0.0; });
    S.lo[i] = 0.0;
    S.hi[i] = 0.0;
  });
  CSDO_FOR(i, 4, { S.cn[i] = 0.0; });
  unsigned act = 0x0780u | 0x1800u | 0x8000u;  // corridor, trust, steer rows exist at every t
  if (t < Nm) {
    act |= ROWS_KIN | ROWS_CTRL;
    const double cst = sincos_of(st).c;
    const double cst2 = cst * cst;  // (steer.cos()).pow(2)
    // x-dyn
    S.c[0][0] = 1.0;
    S.c[0][1] = -dt * (v * sy);
    S.c[0][2] = dt * cy;
    S.cn[0] = -1.0;
    S.lo[0] = S.hi[0] = -(dt * yaw * v * sy);
    // y-dyn
    S.c[1][0] = 1.0;
    S.c[1][1] = dt * (v * cy);
    S.c[1][2] = dt * sy;
    S.cn[1] = -1.0;
    S.lo[1] = S.hi[1] = -(-dt * yaw * v * cy);
    // yaw-dyn
    S.c[2][0] = 1.0;
    S.c[2][1] = (dt / WB * v) / cst2;
    S.c[2][2] = dt / WB * tan_of(st);
    S.cn[2] = -1.0;
    S.lo[2] = S.hi[2] = -(-dt * (st * v / WB / cst2));
    // steer-dyn
    S.c[3][0] = 1.0;
    S.c[3][1] = dt * 1.0;
    S.cn[3] = -1.0;
    S.lo[3] = S.hi[3] = -0.0;
    // control boxes
    S.c[13][0] = 1.0;
    S.lo[13] = -P.max_v;
    S.hi[13] = P.max_v;
    S.c[14][0] = 1.0;
    S.lo[14] = -P.max_omega;
    S.hi[14] = P.max_omega;
  }
  if (t == 0 || t == Nm) {  // start / goal pose pinned to the ORIGINAL initial guess
    act |= ROWS_CFG;
    S.c[4][0] = 1.0;
    S.lo[4] = S.hi[4] = xt;
    S.c[5][0] = 1.0;
    S.lo[5] = S.hi[5] = yt;
    S.c[6][0] = 1.0;
    S.lo[6] = S.hi[6] = yawt;
  }
  // corridor rows: disc-centre linearisation
  dyaw_f_x = -P.f2x * sy;
  dyaw_f_y = P.f2x * cy;
  dyaw_r_x = -P.r2x * sy;
  dyaw_r_y = P.r2x * cy;
  e_xf = P.f2x * (cy + yaw * sy);
  e_yf = P.f2x * (sy - yaw * cy);
  e_xr = P.r2x * (cy + yaw * sy);
  e_yr = P.r2x * (sy - yaw * cy);
  S.c[7][0] = 1.0;  S.c[7][1] = dyaw_f_x;  S.lo[7] = CD(C_CLB + 0, t) - e_xf;  S.hi[7] = CD(C_CUB + 0, t) - e_xf;
  S.c[8][0] = 1.0;  S.c[8][1] = dyaw_f_y;  S.lo[8] = CD(C_CLB + 1, t) - e_yf;  S.hi[8] = CD(C_CUB + 1, t) - e_yf;
  S.c[9][0] = 1.0;  S.c[9][1] = dyaw_r_x;  S.lo[9] = CD(C_CLB + 2, t) - e_xr;  S.hi[9] = CD(C_CUB + 2, t) - e_xr;
  S.c[10][0] = 1.0; S.c[10][1] = dyaw_r_y; S.lo[10] = CD(C_CLB + 3, t) - e_yr; S.hi[10] = CD(C_CUB + 3, t) - e_yr;
  // trust region around the original guess
  S.c[11][0] = 1.0; S.lo[11] = -P.r_trust + xt; S.hi[11] = P.r_trust + xt;
  S.c[12][0] = 1.0; S.lo[12] = -P.r_trust + yt; S.hi[12] = P.r_trust + yt;
  // steer box
  const double steer_max = P.steer_max;   // atan(WB / r), formed by the host (batch_pack.h: make_params)
  S.c[15][0] = 1.0; S.lo[15] = -steer_max; S.hi[15] = steer_max;
  S.act = act;
  // objective: first-difference Laplacian on v (Neumann ends), identity on w
  S.Pvv = (t < Nm) ? ((t == 0 || t == Nt - 2) ? 1.0 : 2.0) : 0.0;
  S.Pvn = (t <= Nt - 3) ? -1.0 : 0.0;
  S.Pww = (t < Nm) ? 1.0 : 0.0;
}

// =========================================================================================================
// BCR factorisation of H = P + sigma I + A' R A, on the solver lanes.
// The diagonal block A_t (packed lower, 21) and the coupling R_t to the node's current right neighbour (36, rows: right
// node's variables, columns: own) live in the workspace between levels (FA coalesced, FR = the node's E_r slot); an
// eliminated node works through its products one at a time so that at most three 6x6 operands are
// live.  The dense tail is inverted in LDS, element-parallel over all solver threads.
// =========================================================================================================
#define FA(k, t) SX(78 + (k), t)
#define FR(k, t) FE(36 + (k), t)
// Pair-split modes: where the solve wants the factor of node t (see "pair-split solve" below).  F_l(t), transposed (entry [c][r]
// at c * 6 + r), goes to the lane that multiplies with it - an odd node's to lane t - 1 (its level-1 block, FE 0..35), an even
// node's to its own lane (the block of its level, FE 72..107); F_r(t), as it is, to the other lane of the pair: lane t (odd node,
// level-1 block) or lane t + 1 (even node).  A node without a right neighbour leaves there the second half of F_l instead
// (columns 3..5, the rest zero): its backward step is two half-sums over F_l, and the partner lane forms the second one.
#define PF_L(idx, t) sh.facE[(unsigned)((t) - ((t) & 1)) * (unsigned)FAC_E_DOUBLES + (unsigned)((((t) & 1) ? 0 : 72) + (idx))]
#define PF_R(idx, t) sh.facE[(unsigned)((t) + 1 - ((t) & 1)) * (unsigned)FAC_E_DOUBLES + (unsigned)((((t) & 1) ? 0 : 72) + (idx))]
#define ROW(r, f) (rows + (int64_t)(f) * rcap)[(unsigned)(r)]
#if defined(CSDO_PROFILE_PHASES)
struct FactorProf {   // diagnostic build: the caller's phase timers (slots 16..18: assembly, levels, tail inversion)
  long long* acc;
  long long* last;
};
#define CSDO_FPHASE(k)                                                 \
  do {                                                                 \
    if (threadIdx.x == 0) {                                            \
      const long long now_ = (long long)__builtin_amdgcn_s_memtime();  \
      fprof.acc[k] += now_ - *fprof.last;                              \
      *fprof.last = now_;                                              \
    }                                                                  \
  } while (0)
#else
struct FactorProf {};
#if defined(CSDO_ASM_MARKS)   // analysis builds only (scripts/asm_serial_loads.py): phase boundaries as comments in the assembly
#define CSDO_FPHASE(k) asm volatile("; CSDO_MARK fphase_" #k)
#else
#define CSDO_FPHASE(k) ((void)0)
#endif
#endif
template <int ROLE, int MODE, bool BIGT>
CSDO_FN void bcr_factor(const Shm& sh_in, const double* rows_in, const int64_t rcap_in, const int32_t* tstart_in,
                              const int Nt_in, const int h_tail_in, const int n_tail_in, const double sigma_in,
                              const double rho_in, const FactorProf fprof, const bool fold_in = false) {
  const Shm& sh = sh_in;
  const double* rows = rows_in;
  const int64_t rcap = rcap_in;
  const int32_t* tstart = tstart_in;
  // h_tail, n_tail: stride and size of the tail the FACTORISATION reduces to and inverts densely.  A folded tail (fold, BIGT only:
  // AgentDesc::tail_nodes = 12) is that of six nodes at twice the solve's stride; its inverse is then EXPANDED by the last level's
  // factors to the explicit inverse of the up to twelve nodes at the solve's stride (the end of this function), and the solve
  // skips that level in both directions.
  const int Nt = Nt_in, Nm = Nt - 1, h_tail = h_tail_in, n_tail = n_tail_in;
#if CSDO_TAIL_BIG >= 2
  const bool fold = BIGT && fold_in;
#else
  constexpr bool fold = false;   // (the shipped build: the folded tail is compiled out, its code below included - measured slower)
  (void)fold_in;
#endif
  const double sigma = sigma_in, rho_now = rho_in;
  // the tail's inverse: row stride and capacity (a constant but for the kernels that read the tail's size at run time, BIGT)
  const int ldt = BIGT ? sh.ld_tinv : (int)LD_tinv;
  const int tcap = BIGT ? (ldt - 2) : (int)TAIL_N;
#define TINV(c, r) sh.tinv[(r) * ldt + (c)]
  CSDO_SYNC();  // the row lanes' workspace writes (set-up stage / save) must be visible to the solver lanes
  CSDO_SLANES(t) {
    const unsigned act = (unsigned)WS(W_ACT, t), eqm = (unsigned)WS(W_EQ, t), lom = (unsigned)WS(W_LOOSE, t);
    double cn4[4];   // (the four loads in flight together)
    CSDO_FOR(k, 4, { cn4[k] = WS(W_CN + k, t); });
    CSDO_FOR(k, 4, {
      const double cnk = cn4[k];
      SH(carry, k, t) = (act & (1u << k)) ? rho_of_masks(eqm, lom, k, rho_now) * cnk * cnk : 0.0;
    });
  }
  CSDO_SYNC();
  CSDO_SLANES(t) {
    // One batch of loads for everything the two blocks need of the home rows (27 coefficients, 4 coupling coefficients, the three
    // objective terms, the masks), and no per-row existence test: a row that does not exist at this t has zero coefficients
    // (assemble_home_rows) and its mask bits are clear, so it adds rho * 0 * 0 = +0.0 to sums that are never -0.0 - the same
    // bits.  (Row by row behind `if (act & bit)` every row's coefficients were a trip to the workspace of their own.)
    const unsigned eqm = (unsigned)WS(W_EQ, t), lom = (unsigned)WS(W_LOOSE, t);
    const int ncols = (t < Nm) ? 6 : 4;
    double cc[NROW][3], cnn[4], pp[3];
    CSDO_FOR(i, NROW, {
      CSDO_FOR(s1, 3, {
        if constexpr (row_col(i, s1) >= 0) cc[i][s1] = WS(W_C + 3 * i + s1, t);
        else cc[i][s1] = 0.0;
      });
    });
    CSDO_FOR(i, 4, { cnn[i] = WS(W_CN + i, t); });
    CSDO_FOR(k, 3, { pp[k] = WS(W_P + k, t); });
    {   // diagonal block
      double A[21];
      CSDO_FOR(k, 21, { A[k] = 0.0; });
      CSDO_FOR(j, 6, { A[sym(j, j)] = (j < ncols) ? sigma : 1.0; });
      A[sym(4, 4)] += pp[0];
      A[sym(5, 5)] += pp[1];
      if (t > 0) CSDO_FOR(k, 4, { A[sym(k, k)] += SH(carry, k, t - 1); });
      CSDO_FOR(i, NROW, {
        const double rh = rho_of_masks(eqm, lom, i, rho_now);
        CSDO_FOR(s1, 3, {
          if constexpr (row_col(i, s1) >= 0) {
            const double rc = rh * cc[i][s1];
            CSDO_FOR(s2, s1 + 1, { A[sym(row_col(i, s1), row_col(i, s2))] = fma(rc, cc[i][s2], A[sym(row_col(i, s1), row_col(i, s2))]); });
          }
        });
      });
      for (int pl_ = tstart[t]; pl_ < tstart[t + 1]; ++pl_) {   // a plane's four rows: their twelve coefficients in flight at once
        double a4[4], b4[4], c4[4];
        CSDO_FOR(q, 4, {
          a4[q] = ROW(4 * pl_ + q, R_CA);
          b4[q] = ROW(4 * pl_ + q, R_CB);
          c4[q] = ROW(4 * pl_ + q, R_CY);
        });
        CSDO_FOR(q, 4, {
          const double a = a4[q], bb = b4[q], cy = c4[q];
          // inter rows have l = -inf and finite u: never loose, never equality (u - l = inf)
          A[sym(0, 0)] = fma(rho_now * a, a, A[sym(0, 0)]);
          A[sym(1, 0)] = fma(rho_now * bb, a, A[sym(1, 0)]);
          A[sym(1, 1)] = fma(rho_now * bb, bb, A[sym(1, 1)]);
          A[sym(2, 0)] = fma(rho_now * cy, a, A[sym(2, 0)]);
          A[sym(2, 1)] = fma(rho_now * cy, bb, A[sym(2, 1)]);
          A[sym(2, 2)] = fma(rho_now * cy, cy, A[sym(2, 2)]);
        });
      }
      CSDO_FOR(k, 21, { FA(k, t) = A[k]; });
      if constexpr (MODE == 0) {   // eliminated at level 1: the pivot inverse now (see the head of the level loop)
        if (h_tail > 1 && (t & 1)) {
          double Sinv[21];
          spd_inverse6(A, Sinv);
          CSDO_FOR(k, 21, {
            WS(W_SINV + k, t) = Sinv[k];
            sh.vec[(size_t)(36 + k) * csdo_opaque_s(sh.stride) + (unsigned)t] = Sinv[k];   // XC(36 + k, t)
          });
        }
      }
    }
    CSDO_STAGE();
    {   // coupling to t+1: kinematic rows (rho c_i c_next_i at [i][col]) and the v_t v_{t+1} term of P
      double R[36];
      CSDO_FOR(k, 36, { R[k] = 0.0; });
      CSDO_FOR(i, 4, {
        const double rh = rho_of_masks(eqm, lom, i, rho_now);
        const double cni = cnn[i];
        CSDO_FOR(s1, 3, {
          if constexpr (row_col(i, s1) >= 0) {
            const double rc = rh * cc[i][s1];
            R[i * 6 + row_col(i, s1)] = fma(rc, cni, R[i * 6 + row_col(i, s1)]);
          }
        });
      });
      R[4 * 6 + 4] += pp[2];
      const bool has_r = (t + 1) < Nt;
      CSDO_FOR(k, 36, { FR(k, t) = has_r ? R[k] : 0.0; });
    }
  }
  CSDO_SYNC();
  CSDO_FPHASE(16);
  for (int h = 1; h < h_tail; h <<= 1) {
    const int m2 = 2 * h - 1;
#define XC(k, tt) sh.vec[(size_t)(k) * csdo_opaque_s(sh.stride) + (unsigned)(tt)]
    if constexpr (MODE == 0) {
      // MODE 0, column-parallel: a level has one eliminated node in every aligned block of 2h lanes and the other lanes of the
      // block are idle, so the node's five 6x6 products (900 FMAs deep on one lane, 13 k cycles per level with their loads) are
      // dealt by COLUMNS to the first G = 2, 3, 6 lanes of the block (h = 1, 2, >= 4): every element is still the same chain of
      // six FMAs - same bits.  The node's own lane inverts the pivot block first and shows it to the others through its LDS
      // column (fields 36..56; 0..35 hold T = Sinv Rl, then V = Sinv Rr', each column owned by the lane that computes it).
      // The exchange of a level goes through LDS: the ADMM block's per-timestep arrays (vec .. fx, 80 doubles per timestep,
      // contiguous) are dead during the factorisation; seen as 80 fields x stride they give every node a column.  The products'
      // Schur complements are DELIVERED into the columns of the two surviving neighbours: fields 0..20 <- U_r of the left
      // neighbour, 21..41 <- U_l of the right neighbour, 42..77 <- the new coupling from the right neighbour.  (Through the
      // workspace the absorption alone was 19 k cycles per level: the workspaces of the 32 agents of an XCD do not fit its L2.)
      // (the node inverted its pivot block where the block was last in its registers: at the end of the assembly - level 1 - or of
      //  the previous level's absorption; no second trip to the workspace for it, one barrier less per level)
      const int G = (h == 1) ? 2 : ((h == 2) ? 3 : 6), per = 6 / G;
      CSDO_STHREADS(l0, nthr) {
        // The loops of this stage are marked cold: by their nesting depth alone the register allocator ranks them above the ADMM
        // iterations and spills the solve's factor entries (122 ms against 73 ms per step).  For the same reason the operands
        // are fetched where they are used (the pivot inverse: broadcast reads from the node's LDS column; the couplings: from the
        // workspace): held in registers across the column loop they are spilled, or push the spills into the solve.
        int l = l0;
        do {   // (one trip unless the block's helper lanes reach past the solver threads)
          const int base = l & ~m2, g = l - base, t = base + h;
          if (g < G && t < Nt) {
            const bool has_r = (t + h) < Nt;
            const size_t st_ = (size_t)csdo_opaque_s(sh.stride);
#define SV(k) sv_[(size_t)(k) * st_]
            // (t, base: taken through an empty asm inside each column's trip - the dozen-odd 32-bit offsets of this block's accesses
            //  (XC at t, base and t + h, the two factor slots, the couplings of two nodes) were otherwise formed in front of the column
            //  loop, spilled, and came back from scratch one by one, each waited for in front of its use: ten round trips per column)
            const int t_o = t, base_o = base;
            {
              int cc = 0;
              do {
                const int t = csdo_keep(t_o), base = csdo_keep(base_o);
                const double* const sv_ = sh.vec + 36 * st_ + (unsigned)t;   // (the pivot inverse's 21 field addresses: formed in the trip that uses them)
                const int c = g * per + cc;
                double rlc[6], tc[6];
                CSDO_FOR(k, 6, { rlc[k] = FR(k * 6 + c, base); });
                CSDO_FOR(r, 6, {             // column c of T = Sinv Rl = F_l, what the solve uses
                  double a = 0.0;
                  CSDO_FOR(k, 6, { a = fma(SV(sym(r, k)), rlc[k], a); });
                  tc[r] = a;
                });
                CSDO_FOR(r, 6, {
                  XC(r * 6 + c, t) = tc[r];
                  PF_L(c * 6 + r, t) = tc[r];
                  if (!has_r) PF_R(c * 6 + r, t) = (c >= 3) ? tc[r] : 0.0;
                });
                CSDO_FOR(ah, 2, {            // column c of U_l = Rl' T (lower triangle), three rows' operands per batch of loads
                  double rl[18];
                  CSDO_FOR(k, 6, { CSDO_FOR(a3, 3, { rl[k * 3 + a3] = FR(k * 6 + 3 * ah + a3, base); }); });
                  CSDO_FOR(a3, 3, {
                    constexpr int a_ = 3 * ah + a3;
                    double a = 0.0;
                    CSDO_FOR(k, 6, { a = fma(rl[k * 3 + a3], tc[k], a); });
                    if (a_ >= c) XC(21 + a_ * (a_ + 1) / 2 + c, base) = a;
                  });
                });
              } while (__builtin_expect(++cc < per, 0));
            }
            if (has_r) {
              int cc = 0;
              do {
                const int t = csdo_keep(t_o), base = csdo_keep(base_o);
                const double* const sv_ = sh.vec + 36 * st_ + (unsigned)t;
                const int c = g * per + cc;
                double tc[6], rrc[6], vc[6];
                CSDO_FOR(k, 6, { tc[k] = XC(k * 6 + c, t); });
                CSDO_FOR(k, 6, { rrc[k] = FR(c * 6 + k, t); });
                CSDO_FOR(r, 6, {             // column c of V = Sinv Rr'; F_r = E_r Sinv = V'
                  double a = 0.0;
                  CSDO_FOR(k, 6, { a = fma(SV(sym(r, k)), rrc[k], a); });
                  vc[r] = a;
                });
                CSDO_FOR(r, 6, { XC(r * 6 + c, t) = vc[r]; });
                CSDO_FOR(ah, 2, {            // three rows of Rr per batch of loads, used by both products:
                  double rr[18];
                  CSDO_FOR(k, 18, { rr[k] = FR(18 * ah + k, t); });
                  CSDO_FOR(a3, 3, {          // column c of the new coupling (right node <- left node) = -Rr T
                    constexpr int a_ = 3 * ah + a3;
                    double a = 0.0;
                    CSDO_FOR(k, 6, { a = fma(rr[a3 * 6 + k], tc[k], a); });
                    XC(42 + a_ * 6 + c, base) = -a;
                  });
                  CSDO_FOR(a3, 3, {          // column c of U_r = Rr V (lower triangle)
                    constexpr int a_ = 3 * ah + a3;
                    double a = 0.0;
                    CSDO_FOR(k, 6, { a = fma(rr[a3 * 6 + k], vc[k], a); });
                    if (a_ >= c) XC(a_ * (a_ + 1) / 2 + c, t + h) = a;
                  });
                });
              } while (__builtin_expect(++cc < per, 0));
            }
#undef SV
          }
          l += nthr;
        } while (__builtin_expect(l < Nt + 6, 0));
      }
    } else
    CSDO_SLANES(t) {  // eliminated nodes
      if ((t & m2) == h) {
        const bool has_r = (t + h) < Nt;
        double Sinv[21];
        {
          double Ain[21];
          CSDO_FOR(k, 21, { Ain[k] = FA(k, t); });
          spd_inverse6(Ain, Sinv);
        }
        CSDO_FOR(k, 21, { WS(W_SINV + k, t) = Sinv[k]; });
        if constexpr (MODE != 3) {
          // Register-lean order: the 6x6 products T = Sinv Rl and V = Sinv Rr' are parked in this lane's stash slot of
          // LDS as they are produced and read back column by column, so that one 6x6 operand, the pivot inverse and a
          // handful of accumulators are all that is live (the all-register version below spills, and a spilled double
          // costs an L2 round trip).  Same products, same summation order, same results.
          // MODE 0: the exchange of a level goes through LDS.  The ADMM block's per-timestep arrays (vec .. fx, 80 doubles per
          // timestep, contiguous) are dead during the factorisation; seen as 80 fields x stride they give every node a
          // column.  An eliminated node parks its products in its own column and DELIVERS its Schur complements into the
          // columns of the two surviving neighbours, which are idle at this level: fields 0..20 <- U_r of the left
          // neighbour, 21..41 <- U_l of the right neighbour, 42..77 <- the new coupling from the right neighbour.  (Through
          // the workspace the absorption alone was 19 k cycles per level: the workspaces of the 32 agents of an XCD do not
          // fit its L2, so the exchange went to HBM and back.)
#define STASH(k) (*((MODE == 0) ? &XC(k, t) : &SH(stash, k, t)))
#define PUT_UL(idx, v) (*((MODE == 0) ? &XC(21 + (idx), t - h) : &SX(idx, t)) = (v))
#define PUT_CPL(idx, v) (*((MODE == 0) ? &XC(42 + (idx), t - h) : &SX(42 + (idx), t)) = (v))
#define PUT_UR(idx, v) (*((MODE == 0) ? &XC(idx, t + h) : &SX(21 + (idx), t)) = (v))
          {
            double Rl[36];
            CSDO_FOR(k, 36, { Rl[k] = FR(k, t - h); });
            CSDO_FOR(r, 6, {
              CSDO_FOR(c, 6, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Sinv[sym(r, k)], Rl[k * 6 + c], a); });
                STASH(r * 6 + c) = a;
                if constexpr (MODE != 3) {  // F_l = Sinv * E_l, what the solve uses (pair-split modes: where its lane reads it)
                  PF_L(c * 6 + r, t) = a;
                  if (!has_r) PF_R(c * 6 + r, t) = (c >= 3) ? a : 0.0;
                } else {
                  FE(r * 6 + c, t) = a;
                }
              });
            });
            CSDO_STAGE();
            CSDO_FOR(b_, 6, {              // U_l = Rl' T, column b of T at a time
              double tc[6];
              CSDO_FOR(k, 6, { tc[k] = STASH(k * 6 + b_); });
              CSDO_FOR(a_, 6, {
                if constexpr (a_ >= b_) {
                  double a = 0.0;
                  CSDO_FOR(k, 6, { a = fma(Rl[k * 6 + a_], tc[k], a); });
                  PUT_UL(sym(a_, b_), a);
                }
              });
            });
          }
          CSDO_STAGE();
          if (has_r) {
            double Rr[36];
            CSDO_FOR(k, 36, { Rr[k] = FR(k, t); });
            CSDO_FOR(b_, 6, {              // new coupling (right node <- left node) = -Rr * T
              double tc[6];
              CSDO_FOR(k, 6, { tc[k] = STASH(k * 6 + b_); });
              CSDO_FOR(a_, 6, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Rr[a_ * 6 + k], tc[k], a); });
                PUT_CPL(a_ * 6 + b_, -a);
              });
            });
            CSDO_STAGE();
            CSDO_FOR(r, 6, {               // V = Sinv * Rr'; F_r = E_r * Sinv = V'
              CSDO_FOR(c, 6, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Sinv[sym(r, k)], Rr[c * 6 + k], a); });
                STASH(r * 6 + c) = a;
              });
            });
            CSDO_STAGE();
            CSDO_FOR(b_, 6, {              // U_r = Rr V, column b of V at a time
              double vc[6];
              CSDO_FOR(k, 6, { vc[k] = STASH(k * 6 + b_); });
              CSDO_FOR(a_, 6, {
                if constexpr (a_ >= b_) {
                  double a = 0.0;
                  CSDO_FOR(k, 6, { a = fma(Rr[a_ * 6 + k], vc[k], a); });
                  PUT_UR(sym(a_, b_), a);
                }
              });
            });
            CSDO_STAGE();
            if constexpr (MODE != 3) CSDO_FOR(r, 6, { CSDO_FOR(c, 6, { PF_R(c * 6 + r, t) = STASH(r * 6 + c); }); });
            else CSDO_FOR(r, 6, { CSDO_FOR(c, 6, { FR(c * 6 + r, t) = STASH(r * 6 + c); }); });
          }
#undef STASH
#undef PUT_UL
#undef PUT_CPL
#undef PUT_UR
        } else {
          // T = Sinv * Rl   (rows: own vars, cols: left node's vars) = F_l, what the solve uses
          double T[36];
          {
            double Rl[36];
            CSDO_FOR(k, 36, { Rl[k] = FR(k, t - h); });
            CSDO_FOR(r, 6, {
              CSDO_FOR(c, 6, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Sinv[sym(r, k)], Rl[k * 6 + c], a); });
                T[r * 6 + c] = a;
                FE(r * 6 + c, t) = a;
              });
            });
            // U_l = Rl' T  -> Schur update of the left neighbour's diagonal block
            CSDO_FOR(a_, 6, {
              CSDO_FOR(b_, a_ + 1, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Rl[k * 6 + a_], T[k * 6 + b_], a); });
                SX(sym(a_, b_), t) = a;
              });
            });
          }
          CSDO_STAGE();
          if (has_r) {
            double Rr[36];
            CSDO_FOR(k, 36, { Rr[k] = FR(k, t); });
            // new coupling (right node <- left node) = -Rr * T
            CSDO_FOR(a_, 6, {
              CSDO_FOR(b_, 6, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Rr[a_ * 6 + k], T[k * 6 + b_], a); });
                SX(42 + a_ * 6 + b_, t) = -a;
              });
            });
            CSDO_STAGE();
            // V = Sinv * Rr'  (rows: own vars, cols: right node's vars); F_r = E_r * Sinv = V'
            double Vm[36];
            CSDO_FOR(r, 6, {
              CSDO_FOR(c, 6, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Sinv[sym(r, k)], Rr[c * 6 + k], a); });
                Vm[r * 6 + c] = a;
              });
            });
            CSDO_FOR(a_, 6, {
              CSDO_FOR(b_, a_ + 1, {
                double a = 0.0;
                CSDO_FOR(k, 6, { a = fma(Rr[a_ * 6 + k], Vm[k * 6 + b_], a); });
                SX(21 + sym(a_, b_), t) = a;
              });
            });
            CSDO_FOR(r, 6, { CSDO_FOR(c, 6, { FR(c * 6 + r, t) = Vm[r * 6 + c]; }); });
          }
        }
      }
    }
    CSDO_SYNC();
    CSDO_FPHASE(19);
    if constexpr (MODE == 0) {
      // The eliminated nodes' lanes (idle otherwise) move what sits in LDS columns to its place in the workspace: their own
      // F_r = V', and - for the surviving node on their left - the coupling to its new right neighbour (fields 42..77 of THAT node's
      // column).  The survivor's lane did that copy itself, in the middle of its work on the diagonal block: five entries of the
      // block sat in scratch while the 36 stores went out and came back one by one behind full waits.  A block of its own, in front
      // of the survivors': they overwrite fields 36..56 of their column with the next level's pivot inverse (node and lane are at
      // most 32 apart and in one aligned group of 64: one wave, program order).
      CSDO_SLANES(t) {
        if ((t & m2) == h) {
          if ((t + h) < Nt) {
            CSDO_FOR(r, 6, { CSDO_FOR(c, 6, { PF_R(c * 6 + r, t) = XC(r * 6 + c, t); }); });
            CSDO_FOR(k, 36, { FR(k, t - h) = XC(42 + k, t - h); });
          } else {
            CSDO_FOR(k, 36, { FR(k, t - h) = 0.0; });
          }
        }
      }
    }
    CSDO_SLANES(t) {  // remaining nodes absorb the Schur complements (and, but for mode 0, take the coupling to their new right neighbour)
      if ((t & m2) == 0) {
        double A[21];
        CSDO_FOR(k, 21, { A[k] = FA(k, t); });
        if constexpr (MODE == 0) {   // delivered into this node's own column (see the elimination above)
          if (t >= h) CSDO_FOR(k, 21, { A[k] -= XC(k, t); });
          if ((t + h) < Nt) {
            CSDO_FOR(k, 21, { A[k] -= XC(21 + k, t); });   // (the coupling to the new right neighbour: copied by the eliminated node's lane, above)
          }
        } else {
          if (t >= h) CSDO_FOR(k, 21, { A[k] -= SX(21 + k, t - h); });
          if ((t + h) < Nt) {
            CSDO_FOR(k, 21, { A[k] -= SX(k, t + h); });
            const bool has_rr = (t + 2 * h) < Nt;
            if (has_rr) {
              CSDO_FOR(k, 36, { FR(k, t) = SX(42 + k, t + h); });
            } else {
              CSDO_FOR(k, 36, { FR(k, t) = 0.0; });
            }
          }
        }
        CSDO_FOR(k, 21, { FA(k, t) = A[k]; });
        if constexpr (MODE == 0) {   // eliminated at the next level: the pivot inverse now, while the block is here (see the head of the level)
          const int h2 = 2 * h;
          if (h2 < h_tail && (t & (2 * h2 - 1)) == h2) {
            double Sinv[21];
            spd_inverse6(A, Sinv);
            CSDO_FOR(k, 21, {
              WS(W_SINV + k, t) = Sinv[k];
              XC(36 + k, t) = Sinv[k];
            });
          }
        }
      }
    }
    CSDO_SYNC();
    CSDO_FPHASE(17);
  }
  CSDO_FPHASE(17);
  // ---- dense tail: the remaining nodes k * h_tail (k < R_tail) form a block-tridiagonal system with diagonal
  // blocks FA and couplings FR.  It is assembled into LDS (sh.tinv, row r = 6k + i) and inverted in place by
  // Gauss-Jordan elimination without pivoting (the matrix is SPD), one element per thread and pivot step; the rows
  // of the inverse stay in LDS for the solves.  Rows / columns >= n_tail are zero.
  CSDO_STHREADS(l, nthr) {
    // (element by element: three elements of a thread at a time with their loads in flight together was measured slower, 61.6
    //  against 61.3 ms - the extra address arithmetic costs more than the trips it saves)
    // (a folded tail: only the corner the six nodes take - the expansion at the end rewrites every element of the inverse)
    const int acap = fold ? (int)TAIL_N : tcap;
    for (int e = l; e < acap * acap; e += nthr) {
      const int r = e / acap, c = e - r * acap;
      const int kn = r / 6, i = r - 6 * kn, jn = kn * h_tail;
      const int kc = c / 6, ic = c - 6 * kc;
      double v = 0.0;
      if (r < n_tail && c < n_tail) {
        if (kc == kn) v = FA((i >= ic) ? (i * (i + 1) / 2 + ic) : (ic * (ic + 1) / 2 + i), jn);
        else if (kc == kn - 1) v = FR(i * 6 + ic, jn - h_tail);   // H(node kn, node kn-1)[i][ic]
        else if (kc == kn + 1) v = FR(ic * 6 + i, jn);            // H(node kn+1, node kn)[ic][i]
      }
      TINV(c, r) = v;
    }
  }
  CSDO_SYNC();
  // In-place inversion by BLOCK Gauss-Jordan with SETS of pivot blocks.  The tail system is block tridiagonal (<= 6 nodes of 6), so
  // its odd nodes are mutually uncoupled, and once they are eliminated so are nodes 0 and 4 of the remaining chain 0 - 2 - 4: three
  // pivot sets {1, 3, 5}, {0, 4}, {2} (the members that exist) instead of six pivots one after the other - the chain of serial 6x6
  // inverses, each on one lane with everything else waiting, is three long instead of six, and a set's inverses are formed side by
  // side.  For a pivot set P with block-diagonal Pinv and everything else R:
  //   A[P,R] <- Pinv A[P,R];   A[R,R] <- A[R,R] - A[R,P] A[P,R];   A[R,P] <- -A[R,P] Pinv;   A[P,P] <- Pinv
  // (round 4: one pivot block per step, 35 k cycles per inversion; scalar Gauss-Jordan before that: 77 k).
  // A tail of up to twelve nodes (BIGT) takes a fourth set: {1, 3, .. 11}, {0, 4, 8}, {2, 10}, {6} - with at most six nodes the
  // first three are the sets above, the fourth is empty.
  double* const pinv = sh.vec;     // 3 (BIGT: 6) x 36 doubles of scratch: the exchange vectors are dead during the factorisation
  const int R_nodes = n_tail / 6;
  for (int grp = 0; grp < (BIGT ? 4 : 3); ++grp) {
    // members of the set: node g0 + k * gs for k < ng
    const int g0 = (grp == 0) ? 1 : ((grp == 1) ? 0 : ((grp == 2) ? 2 : 6)), gsh = (grp == 0) ? 1 : ((grp == 1 || !BIGT) ? 2 : 3),
              gs = 1 << gsh;   // (stride 2, 4 or 8: shifts, no division)
    const int ng = (g0 < R_nodes) ? (((R_nodes - 1 - g0) >> gsh) + 1) : 0;
    if (ng == 0) continue;
    auto member = [&](const int node) __attribute__((always_inline)) -> int {   // index of the node in the set, or -1
      const int d = node - g0;
      return (d >= 0 && (d & (gs - 1)) == 0 && (d >> gsh) < ng) ? (d >> gsh) : -1;
    };
    auto node_of = [&](const int rc) __attribute__((always_inline)) -> int {   // rc / 6 for rc < 72
      return (rc * 43) >> 8;
    };
    CSDO_TLANES(t) {   // the set's pivot inverses, one lane each: packed lower triangle -> full 6x6 inverse
      if (t < ng) {
        const int p = g0 + t * gs;
        double Ain[21], Pin[21];
        CSDO_FOR(r, 6, { CSDO_FOR(c, r + 1, { Ain[sym(r, c)] = TINV(6 * p + c, 6 * p + r); }); });
        spd_inverse6(Ain, Pin);
        CSDO_FOR(r, 6, { CSDO_FOR(c, 6, { pinv[36 * t + r * 6 + c] = Pin[sym(r, c)]; }); });
      }
    }
    CSDO_SYNC();
    CSDO_STHREADS(l, nthr) {   // row blocks: one thread per (member, column outside the set)
      for (int e = l; e < ng * n_tail; e += nthr) {
        const int k = (e >= n_tail) + (e >= 2 * n_tail) + (BIGT ? ((e >= 3 * n_tail) + (e >= 4 * n_tail) + (e >= 5 * n_tail)) : 0), c = e - k * n_tail;
        if (member(node_of(c)) < 0) {
          const int p0 = 6 * (g0 + k * gs);
          double a[6], nw[6];
          CSDO_FOR(j, 6, { a[j] = TINV(c, p0 + j); });
          CSDO_FOR(i, 6, {
            double v = 0.0;
            CSDO_FOR(j, 6, { v = fma(pinv[36 * k + i * 6 + j], a[j], v); });
            nw[i] = v;
          });
          CSDO_FOR(i, 6, { TINV(c, p0 + i) = nw[i]; });
        }
      }
    }
    CSDO_SYNC();
    CSDO_STHREADS(l, nthr) {   // everything outside the set's rows and columns: minus (its columns) x (the set's new rows), member by member
      for (int e = l; e < n_tail * n_tail; e += nthr) {
        const int r = e / n_tail, c = e - r * n_tail;   // (by a float reciprocal instead: 57.49 against 57.07 ms - the allocation, not the division)
        if (member(node_of(r)) < 0 && member(node_of(c)) < 0) {
          double v = TINV(c, r);
          for (int k = 0; k < ng; ++k) {
            const int p0 = 6 * (g0 + k * gs);
            CSDO_FOR(j, 6, { v = fma(-TINV(p0 + j, r), TINV(c, p0 + j), v); });
          }
          TINV(c, r) = v;
        }
      }
    }
    CSDO_SYNC();
    CSDO_STHREADS(l, nthr) {   // column blocks, and the pivot blocks themselves (the blocks between two members are zero and stay zero)
      for (int e = l; e < ng * n_tail; e += nthr) {
        const int k = (e >= n_tail) + (e >= 2 * n_tail) + (BIGT ? ((e >= 3 * n_tail) + (e >= 4 * n_tail) + (e >= 5 * n_tail)) : 0), r = e - k * n_tail;
        const int p0 = 6 * (g0 + k * gs);
        const int mr = member(node_of(r));
        if (mr < 0) {
          double a[6], nw[6];
          CSDO_FOR(j, 6, { a[j] = TINV(p0 + j, r); });
          CSDO_FOR(i, 6, {
            double v = 0.0;
            CSDO_FOR(j, 6, { v = fma(-a[j], pinv[36 * k + j * 6 + i], v); });
            nw[i] = v;
          });
          CSDO_FOR(i, 6, { TINV(p0 + i, r) = nw[i]; });
        } else if (mr == k) {
          CSDO_FOR(i, 6, { TINV(p0 + i, r) = pinv[36 * k + (r - p0) * 6 + i]; });
        }
      }
    }
    CSDO_SYNC();
  }
#if CSDO_TAIL_BIG >= 2
  if constexpr (BIGT && MODE != 3) {
    if (fold) {
      // ---- the folded last level.  G6 = inverse of the system of the nodes a * 2hs (hs = h_tail / 2: the solve's stride), in the top
      // left corner of TINV.  The nodes o = (2 a + 1) hs in between were eliminated at the last level with
      //   x_o = Sinv_o b_o - T_o x_left - V_o x_right,   b_left -= T_o' b_o,   b_right -= V_o' b_o      (T = Sinv Rl, V = Sinv Rr'),
      // so the inverse of the system of ALL nodes k hs (k < 12) is, block by block (e, e' even nodes, o, o' odd ones):
      //   G[e, e'] = G6[e, e']                       G[e, o] = -(G6[e, l(o)] T_o' + G6[e, r(o)] V_o') = G[o, e]'
      //   G[o, o'] = [o = o'] Sinv_o - T_o G[l(o), o'] - V_o G[r(o), o']
      // - 2 x 36 x 36 elements of twelve multiply-adds, two passes over all solver threads, instead of a level of the reduction in
      // every solve (forward and backward: a tenth of an ADMM iteration) or the dense inversion of a 72 x 72 system (round 5:
      // more than the level gives back).  Scratch: the ADMM block's LDS arrays, dead during the factorisation (G6's copy, then T, V
      // and Sinv of the odd nodes - from the pair-split solve's factor slots in the workspace, where the level left them).
      const int hs = h_tail >> 1;
      double* const s6 = sh.vec + 256;          // [36][36]
      double* const tv = s6 + 36 * 36;           // per odd node a: T [j][k] at 108 a, V [j][k] at 108 a + 36, Sinv [i][j] at 108 a + 72
      CSDO_STHREADS(l, nthr) {
        for (int e = l; e < 36 * 36; e += nthr) {
          const int r = (e * 1821) >> 16, c = e - 36 * r;   // e / 36 for e < 1296
          s6[e] = (r < n_tail && c < n_tail) ? TINV(c, r) : 0.0;
        }
        for (int e = l; e < 6 * 108; e += nthr) {
          const int a = (e * 607) >> 16, idx = e - 108 * a;   // e / 108 for e < 648
          const int to = (2 * a + 1) * hs;
          double v = 0.0;
          if (to < Nt) {
            if (idx < 36) {
              const int j = idx / 6, k = idx - 6 * j;
              v = PF_L(k * 6 + j, to);
            } else if (idx < 72) {
              const int j = (idx - 36) / 6, k = idx - 36 - 6 * j;
              v = ((to + hs) < Nt) ? PF_R(k * 6 + j, to) : 0.0;
            } else {
              const int i = (idx - 72) / 6, j = idx - 72 - 6 * i;
              v = WS(W_SINV + ((i >= j) ? (i * (i + 1) / 2 + j) : (j * (j + 1) / 2 + i)), to);
            }
          }
          tv[e] = v;
        }
      }
      CSDO_SYNC();
      CSDO_STHREADS(l, nthr) {   // the even-even blocks (G6 scattered) and the even-odd blocks with their transposes: every element, zeros included
        for (int e = l; e < 2 * 36 * 36; e += nthr) {
          const bool eo = e >= 36 * 36;
          const int e1 = eo ? e - 36 * 36 : e;
          const int r = (e1 * 1821) >> 16, c = e1 - 36 * r;
          const int ra = (r * 43) >> 8, ri = r - 6 * ra, ca = (c * 43) >> 8, cj = c - 6 * ca;
          const int row = 12 * ra + ri;            // even node ra = node 2 ra of the twelve
          if (!eo) {
            TINV(12 * ca + cj, row) = s6[36 * r + c];
          } else {
            // column: component cj of odd node ca (node 2 ca + 1 of the twelve); its neighbours among the even nodes: ca and ca + 1
            const double* const To = tv + 108 * ca + 6 * cj;
            const double* const Vo = To + 36;
            const double* const gl = s6 + 36 * r + 6 * ca;
            const int car = (ca + 1 < 6) ? ca + 1 : ca;   // (no right neighbour: V is zero)
            const double* const gr = s6 + 36 * r + 6 * car;
            double tl_[6], vr_[6], g1[6], g2[6];
            CSDO_FOR(k, 6, {
              tl_[k] = To[k];
              vr_[k] = Vo[k];
              g1[k] = gl[k];
              g2[k] = gr[k];
            });
            double v = 0.0;
            CSDO_FOR(k, 6, { v = fma(g1[k], tl_[k], v); });
            CSDO_FOR(k, 6, { v = fma(g2[k], vr_[k], v); });
            const int col = 12 * ca + 6 + cj;
            TINV(col, row) = 0.0 - v;
            TINV(row, col) = 0.0 - v;
          }
        }
      }
      CSDO_SYNC();
      CSDO_STHREADS(l, nthr) {   // the odd-odd blocks (zero where a node does not exist: T, V and Sinv are)
        for (int e = l; e < 36 * 36; e += nthr) {
          const int r = (e * 1821) >> 16, c = e - 36 * r;
          const int ra = (r * 43) >> 8, ri = r - 6 * ra, ca = (c * 43) >> 8, cj = c - 6 * ca;
          const int row = 12 * ra + 6 + ri, col = 12 * ca + 6 + cj;
          const double* const To = tv + 108 * ra + 6 * ri;
          const double* const Vo = To + 36;
          const int rar = (ra + 1 < 6) ? ra + 1 : ra;
          double tl_[6], vr_[6], g1[6], g2[6];
          CSDO_FOR(k, 6, {
            tl_[k] = To[k];
            vr_[k] = Vo[k];
            g1[k] = TINV(col, 12 * ra + k);
            g2[k] = TINV(col, 12 * rar + k);
          });
          double v = (ra == ca) ? tv[108 * ra + 72 + 6 * ri + cj] : 0.0;
          CSDO_FOR(k, 6, { v = fma(-tl_[k], g1[k], v); });
          CSDO_FOR(k, 6, { v = fma(-vr_[k], g2[k], v); });
          TINV(col, row) = v;
        }
      }
      CSDO_SYNC();
    }
  }
#endif
  if constexpr (MODE == 3) {   // (node 0 is never eliminated: its lane's blocks are loaded with the others and never used)
    CSDO_TLANES(t) {
      if (t == 0) CSDO_FOR(k, 72, { FE(k, 0) = 0.0; });
    }
  }
  CSDO_SYNC();
  CSDO_FPHASE(18);
}
#undef FA
#undef FR
#undef PF_L
#undef PF_R
#undef XC
#undef ROW
#undef TINV

// =========================================================================================================
// REFINE (csdo_qp_parm::solve_refinement; separate kernel instantiations, compiled out of the default ones): every ADMM iteration's
// linear solve is followed by ONE step of iterative refinement on the residual of the KKT system - see "refinement" in the iteration.
// REFINE: 0 off, 1 a second solve on the residual in every iteration, 2 LAGGED - the residual is formed but not solved for: it joins
// the next iteration's rhs (one solve per iteration).
template <int ROLE, int MODE, bool BIGT, int REFINE = 0, class RowStore, class SolvStore>
CSDO_FN void agent_program(const DeviceBatch& B, const int agent, const Shm& sh, RowStore&& lanes_r,
                           SolvStore&& lanes_s, ProgramOut& out) {
  AgentDesc ad = B.agents[agent];
  ad.Nt = uniform_i32(ad.Nt);
  ad.world = uniform_i32(ad.world);
  ad.n_planes = uniform_i32(ad.n_planes);
  ad.x0_off = uniform_i64(ad.x0_off);
  ad.plane_off = uniform_i64(ad.plane_off);
  ad.tstart_off = uniform_i64(ad.tstart_off);
  ad.rows_off = uniform_i64(ad.rows_off);
  ad.fac_off = uniform_i64(ad.fac_off);
  ad.out_off = uniform_i64(ad.out_off);
  WorldDesc wd = B.worlds[ad.world];
  wd.dimx = uniform_f64(wd.dimx);
  wd.dimy = uniform_f64(wd.dimy);
  wd.obs_off = uniform_i32(wd.obs_off);
  wd.n_obs = uniform_i32(wd.n_obs);
  const SolverParams& P = B.prm;
  const int Nt = ad.Nt, Nm = Nt - 1;
  const double* x0g = B.x0 + ad.x0_off;
  const PlaneDev* planes = B.planes + ad.plane_off;
  const int32_t* tstart = B.tstart + ad.tstart_off;
  double* rows = B.rows_ws + ad.rows_off * ROWS_WS_STRIDE;
  const int64_t rcap = (int64_t)4 * ad.n_planes;
  const int n_obs = wd.n_obs;
  const double dimx = wd.dimx, dimy = wd.dimy, rv = P.rv;
  const bool has_inter = ad.n_planes > 0;
  const double sigma = P.sigma, alpha = P.alpha;
  const int n_vars = 6 * Nt - 2;
  int h_tail = 1;                       // BCR levels run for h < h_tail; nodes k * h_tail form the dense tail
  // (BIGT kernels: the tail's capacity is the agent's, AgentDesc::tail_nodes - 6, 8 or 12, see dsqp_class.h)
  const int tail_cap = BIGT ? ((uniform_i32(ad.tail_nodes) > TAIL_NODES) ? uniform_i32(ad.tail_nodes) : (int)TAIL_NODES) : (int)TAIL_NODES;
  const int ldt = BIGT ? sh.ld_tinv : (int)LD_tinv;      // row stride of the tail's inverse
  const int tvh = BIGT ? sh.tvec_half : (int)TAIL_N;     // offset of tvec's second half
  (void)ldt;
#define TINV(c, r) sh.tinv[(r) * ldt + (c)]
  while ((Nt + h_tail - 1) / h_tail > tail_cap) h_tail <<= 1;
  const int R_tail = (Nt + h_tail - 1) / h_tail, n_tail = 6 * R_tail;
  // A twelve-node tail is FOLDED (bcr_factor): the factorisation reduces to the six nodes at twice the stride, inverts those
  // densely and expands the inverse by the last level's factors; the solve runs on (h_tail, n_tail) as with any other tail.
#if CSDO_TAIL_BIG >= 2
  const bool tail_fold = BIGT && MODE != 3 && tail_cap > 8;   // (mode 3, the one-lane form: the dense inversion of all twelve)
#else
  constexpr bool tail_fold = false;
#endif
  const int h_fac = tail_fold ? 2 * h_tail : h_tail, n_fac = tail_fold ? 6 * ((Nt + h_fac - 1) / h_fac) : n_tail;
  int lg_tail = 0;
  while ((1 << lg_tail) < h_tail) ++lg_tail;
  (void)lg_tail;
  const int NtE = (Nt + 1) & ~1;        // lanes of the pair-split solve: the horizon rounded up to whole pairs
  (void)NtE;
#if defined(CSDO_PROFILE_PHASES)
  long long prof_acc[24] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // 16..23: factor sub-phases
  long long prof_last = (long long)__builtin_amdgcn_s_memtime();
  int prof_cur = 0;
  long long sub_acc[6] = {0, 0, 0, 0, 0, 0}, sub_last = 0;   // free sub-timers of whatever is under the microscope: prof[40..45]
#define CSDO_SUB_RESET() do { if (threadIdx.x == 0) sub_last = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define CSDO_SUB(k) do { if (threadIdx.x == 0) { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); sub_acc[k] += n_ - sub_last; sub_last = n_; } } while (0)
  long long lvl_fwd = 0, lvl_bwd = 0;  // solver lane t == h: cycles inside its own elimination block
  // pair-split solve: a stopwatch every wave keeps for itself (scalar registers); the first and the third solver wave report
  long long xs_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, xs_last = 0;
#define CSDO_XT_RESET() xs_last = (long long)__builtin_amdgcn_s_memtime()
#define CSDO_XT(k) do { const long long n_ = (long long)__builtin_amdgcn_s_memtime(); xs_acc[k] += n_ - xs_last; xs_last = n_; } while (0)
#define CSDO_LVL_BEGIN() const long long lvl_t0 = (long long)__builtin_amdgcn_s_memtime()
#define CSDO_LVL_END(acc) acc += (long long)__builtin_amdgcn_s_memtime() - lvl_t0
#else
#define CSDO_LVL_BEGIN() ((void)0)
#define CSDO_LVL_END(acc) ((void)0)
#define CSDO_XT_RESET() ((void)0)
#define CSDO_XT(k) ((void)0)
#define CSDO_SUB_RESET() ((void)0)
#if defined(CSDO_ASM_MARKS)
#define CSDO_SUB(k) asm volatile("; CSDO_MARK sub_" #k)
#else
#define CSDO_SUB(k) ((void)0)
#endif
#endif

#define ROW(r, f) (rows + (int64_t)(f) * rcap)[(unsigned)(r)]

  // ---------------------------------------------------------------- phase 0: stage obstacles, load the guess
  CSDO_LANES(t) {
    for (int k = t; k < n_obs; k += Nt) {
      const double* o = B.obstacles + (int64_t)(wd.obs_off + k) * 3;
      sh.obs[k] = o[0];
      sh.obs[n_obs + k] = o[1];
      sh.obs[2 * n_obs + k] = o[2] + rv;   // every user needs the obstacle radius inflated by the disc radius
    }
    LaneState& S = CSDO_LS(t);
    CSDO_FOR(k, 6, {
      const double v = (t == Nm && k >= 4) ? 0.0 : x0g[(int64_t)t * 6 + k];
      CD(C_SOL0 + k, t) = v;
      CD(C_SOL + k, t) = v;
    });
    CD(C_XT, t) = x0g[(int64_t)t * 6 + 0];
    CD(C_YT, t) = x0g[(int64_t)t * 6 + 1];
    CD(C_YAWT, t) = x0g[(int64_t)t * 6 + 2];
    S.ncols = (t < Nm) ? 6 : 4;
    S.eqmask = 0;
    S.loosemask = 0;
    if (t < 2 * tvh) sh.tvec[t] = 0.0;
    if (t == 0 && Nt < 2 * tvh)
      for (int k = Nt; k < 2 * tvh; ++k) sh.tvec[k] = 0.0;
  }
  CSDO_SYNC();

  constexpr int n_block_fields = (MODE == 3 ? 30 : (MODE == 2 ? LD_block2 : (MODE == 1 ? LD_block1 : LD_block)));
  // grow_box's per-obstacle step counts (BoxCache): the ADMM block's LDS arrays are idle whenever boxes are grown
#if defined(CSDO_LANE_MODE_DEVICE)
  // (all but the last two fields of them: those carry the initial boxes' status flags, written while other lanes still grow)
  const int box_cap_ = (int)(((size_t)(n_block_fields - 2) * (size_t)sh.stride * 2) / (size_t)blockDim.x);
  const BoxCache box_cache{(unsigned*)sh.vec + CSDO_TID, (int)blockDim.x, box_cap_ < 32 ? box_cap_ : 32};
#else
  unsigned box_words_[32];
  const BoxCache box_cache{box_words_, 1, 32};
#endif
  CSDO_PHASE(1);
  // ---------------------------------------------------------------- initial corridors (calcCorridors :164-248)
  // State's disc centres are float members (motion_planning.h:115-118,229-230): round through float here.
  CSDO_SLANES(t) {  // rear disc on the solver lane
    const double px = CD(C_SOL0 + 0, t), py = CD(C_SOL0 + 1, t), pyaw = CD(C_SOL0 + 2, t);
    const SinCos scp_ = sincos_of(pyaw);
    const double spy = scp_.s, cpy = scp_.c;
    const double xr = (double)(float)(px + P.r2x * cpy), yr = (double)(float)(py + P.r2x * spy);
    BoxD br;
    const int sr = box_at(xr, yr, sh.obs, n_obs, dimx, dimy, rv, br, box_cache);
    CD(C_CLB + 2, t) = br.x_min; CD(C_CLB + 3, t) = br.y_min;
    CD(C_CUB + 2, t) = br.x_max; CD(C_CUB + 3, t) = br.y_max;
    SU(n_block_fields - 1, t) = ((sr >> 1) > 0) ? 1.0 : 0.0;
  }
  double my_flag = 0.0;
  CSDO_LANES(t) {   // front disc on the row lane
    const double px = CD(C_SOL0 + 0, t), py = CD(C_SOL0 + 1, t), pyaw = CD(C_SOL0 + 2, t);
    const SinCos scp_ = sincos_of(pyaw);
    const double spy = scp_.s, cpy = scp_.c;
    const double xf = (double)(float)(px + P.f2x * cpy), yf = (double)(float)(py + P.f2x * spy);
    BoxD bf;
    const int sf = box_at(xf, yf, sh.obs, n_obs, dimx, dimy, rv, bf, box_cache);
    CD(C_CLB + 0, t) = bf.x_min; CD(C_CLB + 1, t) = bf.y_min;
    CD(C_CUB + 0, t) = bf.x_max; CD(C_CUB + 1, t) = bf.y_max;
    SU(n_block_fields - 2, t) = ((sf >> 1) > 0) ? 1.0 : 0.0;
  }
  (void)my_flag;
  CSDO_SYNC();
  CSDO_LANES(t) {
    const double part[1] = {dmax(SU(n_block_fields - 2, t), SU(n_block_fields - 1, t))};
    red_put<1>(sh, t, part);
  }
  {
    double r[1];
    red_fold<1, false>(sh, Nt, r);
    out.static_legal = (r[0] > 0.0) ? 0 : 1;
  }

  const double th = P.delta_solution_threshold;
  double delta = th + 1.0;
  int it = 0, status = 1, admm_total = 0;

#if defined(CSDO_ABL_FIXED)
  while (it < P.max_iter) {
#else
  while (delta > th && it < P.max_iter) {
#endif
  CSDO_PHASE(2);
    // ============================================================== assemble the QP (unscaled)
    CSDO_MARK("assemble");
    // Set-up stage storage.  SU(k, t): field-major scratch over the ADMM block's exchange arrays, which are idle here (every
    // residency mode has >= 30 fields): 0..3 |cn| and 4 |Pvn| for t+1; 5..9 the column factors D_t[0..4] of the pass (for t-1
    // and for the inter-vehicle rows), after the last pass the scaled x of the warm start; 10..12 column maxima of the
    // timestep's inter-vehicle rows; 13 the cost-scaling partial; 14..29 the accumulated row scalings E (14..21 first carry the
    // disc linearisation of the timestep to the threads that assemble the inter-vehicle rows).
    // The inter-vehicle rows (a, b, c_yaw, E, u per row, the timestep per plane) live in the `fx` part of the block's arrays
    // whenever they fit (fields x 4K rows), else in the workspace, and ALL work on them is done row by row by the solver
    // threads, beside the row lanes: assembled and scaled by the rows of a timestep's lane they were a serial chain with a
    // reciprocal square root per row and pass (and, in the workspace, a trip to HBM per access): 30 k cycles per pass.
    // Column maxima reach the timestep's lane through an LDS maximum (order independent), its column factors come back
    // through SU: same operations on the same operands as OSQP's scale_data, same results.
    const bool rz_lds = (MODE == 0) && ((int64_t)21 * ad.n_planes <= (int64_t)(LD_block - 24 - LD_lohi) * sh.stride);
    // (two instantiations of every loop over the rows, chosen per agent: one accessor that picks LDS or the workspace per access
    // compiles to FLAT loads and stores through selected 64-bit addresses)
    double* const fx_lds = (MODE == 0) ? lds_ptr(sh.fx) : nullptr;
#define RZ(r, f_lds, f_ws) (*(L ? &fx_lds[(f_lds) * csdo_opaque_s((int)rcap) + (r)] : &ROW(r, f_ws)))   /* f_lds: a, b, c_yaw, E, u */
#define RZ_TIME(pl_) (L ? (int)fx_lds[5 * csdo_opaque_s((int)rcap) + (pl_)] : (int)planes[pl_].t)
#define DACC(j, t) (*((MODE == 3) ? &CD(C_D + (j), t) : &SU(30 + (j), t)))
    CSDO_LANES(t) {
      LaneState& S = CSDO_LS(t);
      double dfx, dfy, drx, dry, exf, eyf, exr, eyr;
      assemble_home_rows(S, sh, t, Nt, P, dfx, dfy, drx, dry, exf, eyf, exr, eyr);
      SU(14, t) = dfx; SU(15, t) = dfy; SU(16, t) = drx; SU(17, t) = dry;
      SU(18, t) = exf; SU(19, t) = eyf; SU(20, t) = exr; SU(21, t) = eyr;
      SU(10, t) = 0.0; SU(11, t) = 0.0; SU(12, t) = 0.0;
      // the accumulated column scaling D: six accumulators less in the registers of the passes.  In LDS (fields 30..35 of the idle
      // block arrays) where the mode has them - read, multiplied and written back where a pass has its column factors, no trip to
      // the workspace and nothing held across the pass -, in the workspace for the lean mode 3 (30 fields); published after the
      // last pass (warm start)
      CSDO_FOR(j, 6, { DACC(j, t) = 1.0; });
      // the (unscaled) bounds wait in the workspace until the warm start: 32 doubles less in the equilibration's registers
      CSDO_FOR(i, NROW, {
        WS(W_LO + i, t) = S.lo[i];
        WS(W_HI + i, t) = S.hi[i];
      });
      if (t == 0) {
        sh.bcast[30] = 0.0;
        sh.bcast[31] = 0.0;
      }
    }
    CSDO_SYNC();

  CSDO_PHASE(3);
    // ============================================================== Ruiz equilibration (scaling.c scale_data)
    CSDO_MARK("ruiz");
    double cscale = 1.0;
    // Nine of the row lane's 38 coefficients wait in LDS between the passes (fields 36..44 of the idle block arrays; CSDO_RUIZ_PARK,
    // modes with that many fields): the single coefficients of the trust, control and steer rows and the corridor rows' yaw
    // coefficients.  The compiler had made the same choice of its own - nine doubles of the lane in scratch across the pass loop,
    // 21 reloads and 9 stores per pass, most of them waited for one by one -; parked by the program they are one batch of LDS reads
    // at the head of a pass and one of writes at its end.  Same operations on the same operands.
    constexpr bool park = (MODE != 3);
    CSDO_LANES(t) {
      LaneState& S = CSDO_LS(t);
      if constexpr (park) {
        CSDO_FOR(i, NROW, {
          CSDO_FOR(s_, 3, {
            if constexpr (ruiz_park_slot(i, s_) >= 0) SU(36 + ruiz_park_slot(i, s_), t) = S.c[i][s_];
          });
        });
      }
    }
    for (int pass = 0; pass < P.scaling_passes; ++pass) {
      CSDO_SUB_RESET();
      CSDO_LANES(t) {  // hand |cn| and |Pvn| to t+1
        LaneState& S = CSDO_LS(t);
        CSDO_FOR(k, 4, { SU(k, t) = fabs(S.cn[k]); });
        SU(4, t) = fabs(S.Pvn);
      }
      CSDO_STHREADS(l, nthr) {   // inter-vehicle rows: (first pass: assemble,) column maxima to the timestep, row factor
        auto body = [&](auto lds_c) __attribute__((always_inline)) {
        constexpr bool L = decltype(lds_c)::value;
        for (int r = l; r < (int)rcap; r += nthr) {
          const int pl_ = r >> 2, q = r & 3;
          double a, bb, cy, e_acc;
          int tp;
          if (pass == 0) {
            // calcInterVehicleConstraint :1097-1129: row = [a, b, a*Dx + b*Dy] on (x,y,yaw)_t, upper bound
            // -(c + (a*Ex + b*Ey)), lower bound -inf
            const PlaneDev& pd = planes[pl_];
            tp = (int)pd.t;
            a = pd.c[3 * q];
            bb = pd.c[3 * q + 1];
            const double cc = pd.c[3 * q + 2];
            const bool front = q < 2;
            const double Dx = front ? SU(14, tp) : SU(16, tp), Dy = front ? SU(15, tp) : SU(17, tp);
            const double Ex = front ? SU(18, tp) : SU(20, tp), Ey = front ? SU(19, tp) : SU(21, tp);
            cy = a * Dx + bb * Dy;
            RZ(r, 4, R_U) = -(cc + ((0.0 + a * Ex) + bb * Ey));
            e_acc = 1.0;
            if (L && q == 0) fx_lds[5 * csdo_opaque_s((int)rcap) + pl_] = (double)tp;
          } else {
            tp = RZ_TIME(pl_);
            a = RZ(r, 0, R_CA);
            bb = RZ(r, 1, R_CB);
            cy = RZ(r, 2, R_CY);
            e_acc = RZ(r, 3, R_E);
          }
          const double fa = fabs(a), fb = fabs(bb), fc = fabs(cy);
          lds_max_nonneg(&SU(10, tp), fa);
          lds_max_nonneg(&SU(11, tp), fb);
          lds_max_nonneg(&SU(12, tp), fc);
          const double rn = nmax(nmax(fa, fb), fc);
          const double et = inv_sqrt_limited(limit_norm(rn));
          RZ(r, 0, R_CA) = a * et;
          RZ(r, 1, R_CB) = bb * et;
          RZ(r, 2, R_CY) = cy * et;
          RZ(r, 3, R_E) = e_acc * et;
        }
        };
        if (rz_lds) body(std::true_type{});
        else body(std::false_type{});
      }
      CSDO_SYNC();
      CSDO_SUB(0);
      CSDO_LANES(t) {
        LaneState& S = CSDO_LS(t);
        double cn_[6] = {0, 0, 0, 0, 0, 0};  // column norms of [P; A]
        double* Dt = S.b;                    // scratch: per-column factor of this pass
        double* Et = S.z;                    // scratch: per-row factor of this pass
        double pk[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // the parked coefficients, for the length of this block
        if constexpr (park) CSDO_FOR(k, 9, { pk[k] = SU(36 + k, t); });
#define RC(i, s_) (*((park && ruiz_park_slot(i, s_) >= 0) ? &pk[ruiz_park_slot(i, s_) >= 0 ? ruiz_park_slot(i, s_) : 0] : &S.c[i][s_]))
        if (t > 0) {
          CSDO_FOR(k, 4, { cn_[k] = SU(k, t - 1); });
          cn_[4] = SU(4, t - 1);
        }
        cn_[4] = nmax(cn_[4], nmax(fabs(S.Pvv), fabs(S.Pvn)));
        cn_[5] = nmax(cn_[5], fabs(S.Pww));
        // Four rows at a time: their norms (with the selects of limit_norm, which are fenced), then the four reciprocal square
        // roots in one stretch of plain arithmetic - four independent chains of 18 dependent fp64 instructions that the
        // scheduler interleaves (row after row they ran at the latency of one chain: a lone wave per SIMD) -, then the factors
        // are applied to the rows' own-column coefficients and folded into E (the column factors follow below, when the
        // column maxima are complete): only the four kinematic rows' factors outlive their group.
        CSDO_FOR(grp, NROW / 4, {
          double ln[4], e4[4], eacc[4];
          CSDO_FOR(q, 4, { eacc[q] = SU(14 + 4 * grp + q, t); });   // (in flight while the group's factors are computed)
          CSDO_FOR(q, 4, {
            constexpr int i = 4 * grp + q;
            // (no test whether the row exists at this t: the coefficients of a row that does not are zero - its norm is 0, its
            //  factor 1, the column maxima unchanged -, and sixteen tests were sixteen basic blocks, each waiting for its own operands)
            double rn = 0.0;
            CSDO_FOR(s, 3, {
              if constexpr (row_col(i, s) >= 0) {
                const double a = fabs(RC(i, s));
                rn = nmax(rn, a);
                cn_[row_col(i, s)] = nmax(cn_[row_col(i, s)], a);
              }
            });
            if constexpr (i < 4) rn = nmax(rn, fabs(S.cn[i]));
            ln[q] = limit_norm(rn);
          });
          CSDO_FOR(q, 4, { e4[q] = inv_sqrt_limited(ln[q]); });
          CSDO_FOR(q, 4, {
            constexpr int i = 4 * grp + q;
            if constexpr (i < 4) Et[i] = e4[q];
            CSDO_FOR(s, 3, {
              if constexpr (row_col(i, s) >= 0) RC(i, s) = RC(i, s) * e4[q];
            });
            SU(14 + i, t) = csdo_one_if(pass == 0, eacc[q]) * e4[q];
          });
        });
        CSDO_FOR(k, 3, {   // the timestep's inter-vehicle rows; cleared for the next pass
          cn_[k] = nmax(cn_[k], SU(10 + k, t));
          SU(10 + k, t) = 0.0;
        });
        {
          double ln[6], r6[6];
          CSDO_FOR(j, 6, { ln[j] = limit_norm(cn_[j]); });
          CSDO_FOR(j, 6, { r6[j] = inv_sqrt_limited(ln[j]); });
          CSDO_FOR(j, 6, { Dt[j] = csdo_one_if(j >= S.ncols, r6[j]); });
        }
        CSDO_FOR(k, 5, { SU(5 + k, t) = Dt[k]; });
        if (t == 0) sh.bcast[30 + ((pass + 1) & 1)] = 0.0;   // the other pass parity's cost-scaling flag
        // everything that only needs the lane's own factors is scaled right here; what needs the right neighbour's column
        // factors (the kinematic rows' coefficients on t+1, P's off-diagonal) waits for the barrier, so that only four row
        // factors and the lane's six column factors stay live across it
        CSDO_FOR(i, NROW, {
          CSDO_FOR(s, 3, {
            if constexpr (row_col(i, s) >= 0) RC(i, s) = RC(i, s) * Dt[row_col(i, s)];
          });
        });
        if constexpr (park) CSDO_FOR(k, 9, { SU(36 + k, t) = pk[k]; });
#undef RC
        S.Pvv = (S.Pvv * Dt[4]) * Dt[4];
        S.Pww = (S.Pww * Dt[5]) * Dt[5];
        {
          double dacc[6];
          CSDO_FOR(j, 6, { dacc[j] = DACC(j, t); });
          CSDO_FOR(j, 6, { DACC(j, t) = dacc[j] * Dt[j]; });
        }
      }
      CSDO_SYNC();
      CSDO_SUB(1);
      CSDO_LANES(t) {
        LaneState& S = CSDO_LS(t);
        const double* Dt = S.b;
        const double* Et = S.z;
        double Dn[5] = {1, 1, 1, 1, 1};
        if (t < Nm) CSDO_FOR(k, 5, { Dn[k] = SU(5 + k, t + 1); });
        CSDO_FOR(i, 4, { S.cn[i] = (S.cn[i] * Et[i]) * Dn[i]; });
        // scaled |P(v_{t-1}, v_t)| as its owner computes it
        double pvn_left = 0.0;
        if (t > 0) pvn_left = (SU(4, t - 1) * SU(9, t - 1)) * Dt[4];
        S.Pvn = (S.Pvn * Dt[4]) * Dn[4];
        // cost normalisation: mean column norm of the scaled P
        double colsum = 0.0;
        if (t < Nm) colsum = nmax(nmax(fabs(S.Pvv), fabs(S.Pvn)), pvn_left) + fabs(S.Pww);
        SU(13, t) = colsum;
        // c_temp = max(sum / n_vars, 1) is 1 - P, c unchanged - unless the sum exceeds n_vars = 6 Nt - 2, which takes a lane
        // above 5: only then is the sum formed (P's columns are equilibrated to ~1 by this very pass; never observed)
        if (colsum > 5.0) sh.bcast[30 + (pass & 1)] = 1.0;
      }
      CSDO_STHREADS(l, nthr) {   // inter-vehicle rows only touch the columns of their own timestep
        auto body = [&](auto lds_c) __attribute__((always_inline)) {
        constexpr bool L = decltype(lds_c)::value;
        for (int r = l; r < (int)rcap; r += nthr) {
          const int tp = RZ_TIME(r >> 2);
          RZ(r, 0, R_CA) = RZ(r, 0, R_CA) * SU(5, tp);
          RZ(r, 1, R_CB) = RZ(r, 1, R_CB) * SU(6, tp);
          RZ(r, 2, R_CY) = RZ(r, 2, R_CY) * SU(7, tp);
        }
        };
        if (rz_lds) body(std::true_type{});
        else body(std::false_type{});
      }
      CSDO_SYNC();
      CSDO_SUB(2);
      if (uniform_f64(sh.bcast[30 + (pass & 1)]) != 0.0) {
        double r[1];
        field_sum(sh, Nt, 13, r);
        double c_temp = r[0] / (double)n_vars;
        c_temp = osqp_max(c_temp, limit_scaling(0.0));  // ||q||_inf = 0 -> 1 (q = 0, :196-197)
        c_temp = limit_scaling(c_temp);
        c_temp = uniform_f64(1.0 / c_temp);
        CSDO_LANES(t) {
          LaneState& S = CSDO_LS(t);
          S.Pvv *= c_temp;
          S.Pww *= c_temp;
          S.Pvn *= c_temp;
        }
        cscale = uniform_f64(cscale * c_temp);
      }
    }
    const double cinv = uniform_f64(1.0 / cscale);
    CSDO_LANES(t) {   // the parked coefficients come back into the lane's registers for the rest of the QP
      LaneState& S = CSDO_LS(t);
      if constexpr (park) {
        CSDO_FOR(i, NROW, {
          CSDO_FOR(s_, 3, {
            if constexpr (ruiz_park_slot(i, s_) >= 0) S.c[i][s_] = SU(36 + ruiz_park_slot(i, s_), t);
          });
        });
      }
    }

  CSDO_PHASE(4);
    // ============================================================== scaled bounds, row classes, warm start
    CSDO_MARK("warmstart");
    CSDO_LANES(t) {
      LaneState& S = CSDO_LS(t);
      unsigned eq = 0, loose = 0;
      // the scaled bounds go straight back to the workspace (the iterations read them from there / from LDS): held in the lane's
      // registers until the publishing loop below they were spilled, and reloaded from scratch one by one in front of their stores
      double lo_u[NROW], hi_u[NROW], e_u[NROW];
      CSDO_FOR(i, NROW, {
        lo_u[i] = WS(W_LO + i, t);
        hi_u[i] = WS(W_HI + i, t);
        e_u[i] = SU(14 + i, t);
      });
      CSDO_FOR(i, NROW, {   // (every row: E = 1 and zero bounds where the row does not exist; classes only for those that do)
        {
          const double Ei = e_u[i];
          CD(C_E + i, t) = Ei;
          const double lo_s = Ei * lo_u[i], hi_s = Ei * hi_u[i];
          WS(W_LO + i, t) = lo_s;
          WS(W_HI + i, t) = hi_s;
          // (without branches: as `if (..) loose |= bit; else if (..) eq |= bit;` the two masks became a two-element array in
          //  scratch, indexed by the outcome, read-modified-written once per row behind a full wait)
          const bool is_loose = lo_s < -OSQP_INFTY * MIN_SCALING && hi_s > OSQP_INFTY * MIN_SCALING;
          const bool is_eq = !is_loose && (hi_s - lo_s < RHO_TOL);
          loose |= is_loose ? (1u << i) : 0u;
          eq |= is_eq ? (1u << i) : 0u;
        }
      });
      eq &= S.act;
      loose &= S.act;
      S.eqmask = eq;
      S.loosemask = loose;
      WS(W_ACT, t) = (double)S.act;
      WS(W_EQ, t) = (double)eq;
      WS(W_LOOSE, t) = (double)loose;
      // osqp_warm_start_x: x <- Dinv x0
      CSDO_FOR(j, 6, {
        const double dj = DACC(j, t);
        if constexpr (MODE != 3) CD(C_D + j, t) = dj;   // (the master copy: read by the residual update and the SQP bookkeeping)
        S.x[j] = (1.0 / dj) * CD(C_SOL0 + j, t);
      });
      CSDO_FOR(k, 4, { SU(5 + k, t) = S.x[k]; });
    }
    CSDO_SYNC();
    CSDO_STHREADS(l, nthr) {   // inter-vehicle rows: z <- A x, y <- 0; the master copy goes to the workspace
      auto body = [&](auto lds_c) __attribute__((always_inline)) {
      constexpr bool L = decltype(lds_c)::value;
      for (int r = l; r < (int)rcap; r += nthr) {
        const int tp = RZ_TIME(r >> 2);
        const double a = RZ(r, 0, R_CA), bb = RZ(r, 1, R_CB), cy = RZ(r, 2, R_CY), e_acc = RZ(r, 3, R_E);
        ROW(r, R_Z) = (a * SU(5, tp) + bb * SU(6, tp)) + cy * SU(7, tp);
        ROW(r, R_Y) = 0.0;
        ROW(r, R_DY) = 0.0;
        const double u_scaled = e_acc * RZ(r, 4, R_U);
        ROW(r, R_U) = u_scaled;
        if (L) {   // publish what the set-up stages kept in LDS
          ROW(r, R_CA) = a;
          ROW(r, R_CB) = bb;
          ROW(r, R_CY) = cy;
          ROW(r, R_E) = e_acc;
        }
      }
      };
      if (rz_lds) body(std::true_type{});
      else body(std::false_type{});
    }
    CSDO_LANES(t) {  // z <- A x, y <- 0
      LaneState& S = CSDO_LS(t);
      double xn[4] = {0, 0, 0, 0};
      if (t < Nm) CSDO_FOR(k, 4, { xn[k] = SU(5 + k, t + 1); });
      double Ax[NROW];
      rows_times_x(S, S.x, xn, Ax);
      CSDO_FOR(i, NROW, {
        S.z[i] = Ax[i];
        S.y[i] = 0.0;
        CD(C_DY + i, t) = 0.0;
      });
      // the set-up stage worked in registers; publish the master copy for the cold phases and for load_hot
      CSDO_FOR(i, NROW, {
        CSDO_FOR(s, 3, {
          if constexpr (row_col(i, s) >= 0) WS(W_C + 3 * i + s, t) = S.c[i][s];
        });
      });
      CSDO_FOR(i, 4, { WS(W_CN + i, t) = S.cn[i]; });
      WS(W_P + 0, t) = S.Pvv;
      WS(W_P + 1, t) = S.Pww;
      WS(W_P + 2, t) = S.Pvn;
    }

    double rho = osqp_min(osqp_max(P.rho0, RHO_MIN), RHO_MAX);

    // BCR factorisation of H = P + sigma I + A' R A: bcr_factor above
    auto factor = [&](const double rho_now) __attribute__((always_inline)) {
      CSDO_MARK("factor_begin");
      CSDO_PHASE(5);
#if defined(CSDO_PROFILE_PHASES)
      const FactorProf fprof{prof_acc, &prof_last};
#else
      const FactorProf fprof{};
#endif
      bcr_factor<ROLE, MODE, BIGT>(sh, rows, rcap, tstart, Nt, h_fac, n_fac, sigma, rho_now, fprof, tail_fold);
    };

    // ============================================================== BCR solve on the solver lanes.
    // In: rhs of node t in V.b (assembled by the solver lane).  Out: x_tilde in V.b and sh.vec[t].
    // Coupling blocks and pivot inverses come from registers: the only LDS traffic is the 6-vectors.
    // ============================================================== pair-split solve (residency modes 0 and 1)
    // A level of the reduction is, per eliminated node, two 6x6 products of the node's rhs (forward: pl = F_l' b for the left
    // neighbour, pr = F_r b for the right one; backward: x = (w - F_l x_left) + (0 - F_r' x_right)) - 72 multiply-adds deep on
    // the node's lane, while at least every other lane of its wave has nothing to do.  Here the two products sit on the two
    // lanes of the node's PAIR (2j, 2j + 1), one block each:
    //   level 1 (odd nodes t):     lane t - 1 holds F_l(t)' and forms pl(t) - which it consumes itself -, lane t holds F_r(t);
    //   levels h >= 2 (even nodes): lane t holds F_l(t)', lane t + 1 (an odd lane: eliminated at level 1, idle since) holds F_r(t).
    // A lane works at level 1 and at ONE further level, so it keeps two blocks (SolvRegs::el, er + Shm::fx).  The rhs reaches the
    // partner lane by a DPP move (quad_perm inside the pair), every element is the same chain of multiply-adds as in the
    // one-lane form (half-sums of three in the forward sweep, chains of six from w and from 0 in the backward one) - bit for bit.
    // Partners of an elimination are h <= 32 lanes apart and, but for one case, in the same WAVE: the levels exchange their
    // partials through LDS without workgroup barriers (a wave's LDS instructions complete in order; level 1 by DPP).  The one
    // case is the first lane of a wave (a multiple of 64: always a tail node), whose LEFT partials come from the last lanes of the
    // wave in front of it: that wave adds them up behind its last level and hands the sum over beside the gathered tail rhs
    // (see the gather); the tail product subtracts it.  Backward, a wave only ever needs x of its own nodes and of the next
    // wave's first node - a tail node, known since the tail product.
    // Barriers per solve: 3 (levels + gather | tail product beside the w pass | backward) instead of 2 log2(Nt / 6) + 2.
    // Measured: fixed-work iteration 10.45 -> 9.3 us on one instance alone, 2.5 % under a full batch.
    constexpr unsigned XF_ABS = 1u, XF_ABSR = 1u << 6, XF_WR = 1u << 12, XF_OWN = 1u << 18, XF_NR = 1u << 24;
    constexpr int XER = (MODE == 2) ? ER_REG2 : ((MODE == 1) ? ER_REG1 : ER_REG), XFX = 36 - XER;   // the second block: entries in registers / elsewhere
#define A2_LDS(k, tt) (*((MODE == 2 && (k) >= LD_fx2) ? &sh.carry[(tt) * LD_carry + 4 + ((k) >= LD_fx2 ? (k) - LD_fx2 : 0)] \
                                                   : &sh.fx[(tt) * (MODE == 2 ? LD_fx2 : (MODE == 1 ? LD_fx1 : LD_fx)) + (k)]))   /* the block's entries beyond XER */
    auto solve_pair = [&](auto before_last_barrier) __attribute__((always_inline)) {   // (generic: only instantiated for the modes that call it)
      CSDO_MARK("solve_begin");
      CSDO_PHASE(7);
#if defined(CSDO_ABL_NOSOLVE)
      CSDO_SYNC();
      return;
#endif
      CSDO_XT(0);   // (since the barrier behind the update: rhs assembly)
#if defined(CSDO_ABL_XNOFWD)
      if (false) {
#else
      if (h_tail > 1) {
#endif
        // (the one-trip loops around the two level-1 steps: the register allocator weighs a value by the loop depth of its uses,
        //  and outside any loop of the iteration the level-1 block would be what it spills)
        CSDO_ONCE_LOOP
        CSDO_XLANES(t) {
          SolvRegs& V = CSDO_SS(t);
          // (768-thread class, 168 registers per lane: the level-1 block is not kept across the iteration - as lane state it was
          //  what the allocator spilled, reloaded entry by entry inside level 1: 6 k cycles against 1 k in the 512-thread class -
          //  but fetched from the workspace in one batch in front of each of its two uses)
          if constexpr (MODE == 2) CSDO_FOR(k, 36, { V.el[k] = FE(k, t < NtE ? t : NtE - 1); });
          CSDO_FOR(k, 6, { V.v[k] = CSDO_XGET(CSDO_DPP_PAIR_ODD, b, k); });   // both lanes of a pair: the odd node's rhs
        CSDO_XSTEP(t)
          {
            double qa[6] = {0, 0, 0, 0, 0, 0}, qb[6] = {0, 0, 0, 0, 0, 0};
            CSDO_FOR(k, 3, {
              CSDO_FOR(a, 6, {
                qa[a] = fma(V.el[a * 6 + k], V.v[k], qa[a]);
                qb[a] = fma(V.el[a * 6 + k + 3], V.v[k + 3], qb[a]);
              });
            });
            CSDO_FOR(a, 6, { V.o[a] = qa[a] + qb[a]; });   // even lane t: pl(t + 1); odd lane t: pr(t)
          }
          if (V.fl & XF_WR) {   // what crosses a wave boundary waits in LDS
            if (t & 1) CSDO_FOR(k, 6, { SH(pr, k, t) = V.o[k]; });
            else CSDO_FOR(k, 6, { SH(pl, k, t + 1) = V.o[k]; });
          }
        CSDO_XSTEP(t)
          double left[6];
          CSDO_FOR(k, 6, { left[k] = CSDO_XGET(CSDO_DPP_PREV, o, k); });
          if (V.fl & XF_ABS) CSDO_FOR(k, 6, { V.b[k] -= left[k]; });
          if (V.fl & XF_ABSR) CSDO_FOR(k, 6, { V.b[k] -= V.o[k]; });
        }
        CSDO_ONCE_END
        CSDO_XT(1);   // forward level 1
        int lev = 1;
        for (int h = 2; h < h_tail; h <<= 1, ++lev) {
          CSDO_XLANES(t) {
            SolvRegs& V = CSDO_SS(t);
            CSDO_FOR(k, 6, { V.v[k] = CSDO_XGET(CSDO_DPP_PAIR_EVEN, b, k); });   // both lanes of a pair: the even node's rhs
          CSDO_XSTEP(t)
            {
              const int tl = t < NtE ? t : NtE - 1;
              double qa[6] = {0, 0, 0, 0, 0, 0}, qb[6] = {0, 0, 0, 0, 0, 0};
#define A2(kk) ((kk) < XER ? V.er[(kk) < XER ? (kk) : 0] : m2x[(kk) >= XER ? (kk) - XER : 0])
              if constexpr (MODE == 2) {
                // (168 registers per lane: two rows of the block at a time - every row is its own pair of accumulation chains -, so
                //  that the part of the block that waits in LDS is never in registers all at once)
                CSDO_FOR(a2, 3, {
                  double rw[12];
                  CSDO_FOR(e, 12, { rw[e] = (a2 * 12 + e) < XER ? V.er[(a2 * 12 + e) < XER ? (a2 * 12 + e) : 0] : A2_LDS((a2 * 12 + e) >= XER ? (a2 * 12 + e) - XER : 0, tl); });
                  CSDO_FOR(k, 3, {
                    CSDO_FOR(ah, 2, {
                      qa[2 * a2 + ah] = fma(rw[ah * 6 + k], V.v[k], qa[2 * a2 + ah]);
                      qb[2 * a2 + ah] = fma(rw[ah * 6 + k + 3], V.v[k + 3], qb[2 * a2 + ah]);
                    });
                  });
                  CSDO_STAGE();
                });
              } else {
                double m2x[XFX > 0 ? XFX : 1];
                CSDO_FOR(k, XFX, { m2x[k] = A2_LDS(k, tl); });
                CSDO_FOR(k, 3, {
                  CSDO_FOR(a, 6, {
                    qa[a] = fma(A2(a * 6 + k), V.v[k], qa[a]);
                    qb[a] = fma(A2(a * 6 + k + 3), V.v[k + 3], qb[a]);
                  });
                });
              }
              CSDO_FOR(a, 6, { V.o[a] = qa[a] + qb[a]; });   // even lane t: pl(t); odd lane t: pr(t - 1)
            }
            if (V.fl & (XF_WR << lev)) {
              if (t & 1) CSDO_FOR(k, 6, { SH(pr, k, t - 1) = V.o[k]; });
              else CSDO_FOR(k, 6, { SH(pl, k, t) = V.o[k]; });
            }
          CSDO_XSTEP_LDS(t)
            {
              // both partials in flight at once: a lane that takes none of the two reads six zeros instead (the hand-over slot of
              // node 0, which has no left neighbours: zero for the whole solve) and subtracts them - b - 0.0 is b - so the wave waits
              // for LDS once per level, not once per side
              const double* const zeros = sh.tvec + tvh;
              const double* const pa = (V.fl & (XF_ABS << lev)) ? &SH(pr, 0, t - h) : zeros;
              const double* const pb = (V.fl & (XF_ABSR << lev)) ? &SH(pl, 0, t + h) : zeros;
              double xa[6], xb[6];
              CSDO_FOR(k, 6, {
                xa[k] = pa[k];
                xb[k] = pb[k];
              });
              CSDO_FOR(k, 6, { V.b[k] -= xa[k]; });
              CSDO_FOR(k, 6, { V.b[k] -= xb[k]; });
            }
          }
        }
      }
      // ---- tail nodes gather their rhs.  The first node of a wave (a multiple of 64: a tail node) takes its LEFT partials from the
      // wave in front of it; that wave sums them up behind its last level - six lanes, one component each, in the order of the
      // levels - and leaves the sum beside the gathered rhs (second half of tvec; zero for every other tail node), the tail product
      // subtracts it.  (With the left partials absorbed one by one in level order, as the one-lane form does, the node's wave
      // has to wait for a barrier in front of the gather and catch up behind it: 2.7 k cycles of every iteration as a loop, 1.1 us
      // of 10.5 even with six lanes and all partials in flight.  The order of those few subtractions is the only thing this
      // form changes in the arithmetic.)
      CSDO_SLANES(t) {
        SolvRegs& V = CSDO_SS(t);
        if ((t & (h_tail - 1)) == 0) {
          const int kn = t >> lg_tail;
          CSDO_FOR(k, 6, { sh.tvec[6 * kn + k] = V.b[k]; });
        }
      }
      if (h_tail > 1) {
#if defined(CSDO_LANE_MODE_DEVICE)
        if constexpr (ROLE != ROLE_ROW) csdo_wave_sync();   // (this wave's own partials of the last level, through LDS)
#endif
        CSDO_HANDOVER_LANES(s, q) {
          double part[6];
          CSDO_FOR(l6, 6, {   // (levels 1, 2, 4, ... 32: one that does not exist reads some valid slot and is skipped below)
            constexpr int h = 1 << l6;
            part[l6] = sh.pr[(s - (h < h_tail ? h : 1)) * LD_pr + q];
          });
          double acc = part[0];
          CSDO_FOR(l6, 5, {
            constexpr int h = 2 << l6;
            if (h < h_tail) acc += part[l6 + 1];
          });
          sh.tvec[tvh + 6 * (s >> lg_tail) + q] = acc;
        }
      }
      CSDO_XT(2);   // forward levels >= 2
      CSDO_PHASE(13);
      CSDO_SYNC();
      CSDO_XT(3);   // wait at the barrier in front of the tail
      CSDO_PHASE(22);
#if defined(CSDO_LANE_MODE_DEVICE)
      // The product of the explicit tail inverse with the gathered rhs runs on the ROW waves (see the long-horizon form below)
#if defined(CSDO_ABL_XNOTAILP)
      if constexpr (false) {
#else
      if constexpr (ROLE == ROLE_ROW) {
#endif
        const int x = CSDO_TID_HOT, wl = x & 63, r = (x >> 6) * 21 + wl / 3, p2 = 2 * (wl % 3);
        // NB blocks of six columns on NW row waves (21 rows of the inverse per wave): 6 on 2 for the six-node tail
        auto product = [&](auto nb_c, auto nw_c) __attribute__((always_inline)) {
          constexpr int NB = decltype(nb_c)::value, NW = decltype(nw_c)::value;
          if (x < 64 * NW) {
            double a0 = 0.0, a1 = 0.0;
            if (wl < 63 && r < n_tail) {
              // (six blocks: all operands in flight at once, as ever; more: four blocks at a time - the row lane's registers are its
              //  rows' state, and 48 more doubles of operands put that state into scratch)
              constexpr int CH = NB <= 6 ? NB : 4;
              CSDO_FOR(c4, NB / CH, {
                double tr0[CH], tr1[CH], tb0[CH], tb1[CH];
                CSDO_FOR(jj, CH, {
                  constexpr int j = c4 * CH + jj;
                  tr0[jj] = TINV(6 * j + p2, r);
                  tr1[jj] = TINV(6 * j + p2 + 1, r);
                  tb0[jj] = sh.tvec[6 * j + p2] - sh.tvec[tvh + 6 * j + p2];
                  tb1[jj] = sh.tvec[6 * j + p2 + 1] - sh.tvec[tvh + 6 * j + p2 + 1];
                });
                CSDO_FOR(jj, CH, {
                  a0 = fma(tr0[jj], tb0[jj], a0);
                  a1 = fma(tr1[jj], tb1[jj], a1);
                });
                if constexpr (NB > 6) CSDO_STAGE();
              });
            }
            const double s01 = a0 + a1;
            const double up1 = wave_next(s01), up2 = wave_next(up1);
            const double s0123 = s01 + up1;
            const double tot = s0123 + up2;
            if (wl < 63 && r < n_tail && p2 == 0) {
              const int kn = r / 6, i = r - 6 * kn;
              sh.vec[(kn * h_tail) * LD_vec + i] = tot;
            }
          }
        };
        if constexpr (!BIGT) {
          product(std::integral_constant<int, 6>{}, std::integral_constant<int, 2>{});
        } else {   // (the blocks beyond the agent's tail are zeros - rows and rhs -: they add +0.0)
          if (tail_cap <= 6) product(std::integral_constant<int, 6>{}, std::integral_constant<int, 2>{});
          else if (tail_cap <= 8) product(std::integral_constant<int, 8>{}, std::integral_constant<int, 3>{});
          else product(std::integral_constant<int, 12>{}, std::integral_constant<int, 4>{});
        }
      }
      if constexpr (ROLE == ROLE_BOTH)
#endif
      CSDO_TLANES_TOP(t) {
        double a4[6] = {0, 0, 0, 0, 0, 0};
        if constexpr (BIGT) {   // (the tail's size is the agent's: block after block of six columns, the same chains)
          for (int c0 = 0; c0 < ldt - 2; c0 += 6)
            CSDO_FOR(i6, 6, { a4[i6] = fma(TINV(c0 + i6, t), sh.tvec[c0 + i6] - sh.tvec[tvh + c0 + i6], a4[i6]); });
        } else
        CSDO_FOR(q, 4, {
          double tr[TAIL_N / 4], tb[TAIL_N / 4];
          CSDO_FOR(c, TAIL_N / 4, {
            tr[c] = TINV(q * (TAIL_N / 4) + c, t);
            tb[c] = sh.tvec[q * (TAIL_N / 4) + c] - sh.tvec[TAIL_N + q * (TAIL_N / 4) + c];
          });
          CSDO_FOR(c, TAIL_N / 4, { a4[(q * (TAIL_N / 4) + c) % 6] = fma(tr[c], tb[c], a4[(q * (TAIL_N / 4) + c) % 6]); });
          CSDO_STAGE();
        });
        const int kn = t / 6, i = t - 6 * kn;
        sh.vec[(kn * h_tail) * LD_vec + i] = ((a4[0] + a4[1]) + (a4[2] + a4[3])) + (a4[4] + a4[5]);
      }
      // beside the tail product: w = Sinv b, the start of every eliminated node's backward chain.  An even node's on its own
      // lane; an odd node's on the even lane to its left, which forms that node's F_l half in the backward sweep.  The odd
      // lanes only form second halves, which start from 0.
      auto sinv_times = [&](const int node, const double (&bb)[6], double (&w6)[6]) __attribute__((always_inline)) {
        const int nd = csdo_keep(node);   // (the LDS address formed here: hoisted out of the iterations it is spilled)
        if constexpr (MODE == 0) {
          // the 21 packed entries in eleven 16-byte reads, once (row by row through sym() they were 36 8-byte reads in two halves)
          double s21[21];
          CSDO_FOR(k, 21, { s21[k] = SINV(k, nd); });
          CSDO_FOR(r, 6, {
            const double s01 = fma(s21[sym(r, 1)], bb[1], s21[sym(r, 0)] * bb[0]);
            const double s23 = fma(s21[sym(r, 3)], bb[3], s21[sym(r, 2)] * bb[2]);
            const double s45 = fma(s21[sym(r, 5)], bb[5], s21[sym(r, 4)] * bb[4]);
            w6[r] = (s01 + s23) + s45;
          });
        } else
        CSDO_FOR(half, 2, {
          double sv[3][6];
          CSDO_FOR(r3, 3, {
            CSDO_FOR(c, 6, { sv[r3][c] = SINV(sym(3 * half + r3, c), nd); });
          });
          CSDO_FOR(r3, 3, {
            const double s01 = fma(sv[r3][1], bb[1], sv[r3][0] * bb[0]);
            const double s23 = fma(sv[r3][3], bb[3], sv[r3][2] * bb[2]);
            const double s45 = fma(sv[r3][5], bb[5], sv[r3][4] * bb[4]);
            w6[3 * half + r3] = (s01 + s23) + s45;
          });
          CSDO_STAGE();
        });
      };
#if defined(CSDO_ABL_XNOW)
      if (false) {
#else
      if (h_tail > 1) {
#endif
        CSDO_XLANES(t) {
          SolvRegs& V = CSDO_SS(t);
          // every eliminated node forms its own w on its own lane (one product per lane, in the shadow of the tail product on
          // the row waves); an odd node's w then moves to the even lane of its pair, where its backward chain starts
          if (t < Nt && ((t & 1) || (t & (h_tail - 1)) != 0)) {
            double w6[6];
            sinv_times(t, V.b, w6);
            CSDO_FOR(k, 6, { V.o[k] = w6[k]; });
          }
        CSDO_XSTEP(t)
          double w1[6];
          CSDO_FOR(k, 6, { w1[k] = CSDO_XGET(CSDO_DPP_PAIR_ODD, o, k); });   // even lane: w of node t + 1
          // the odd lanes only form second half-sums, which start from 0; an even node's chain starts from its w (tail nodes keep
          // their rhs, which is not used again); w of the odd node waits in the even lane's own slot of the forward sweep's
          // partials, idle by now, for the last backward level (six doubles less in registers across the sweep)
          if (t & 1) {
            CSDO_FOR(k, 6, { V.b[k] = 0.0; });
          } else {
            if (t < Nt && (t & (h_tail - 1)) != 0) CSDO_FOR(k, 6, { V.b[k] = V.o[k]; });
            if ((t + 1) < Nt) CSDO_FOR(k, 6, { SH(pr, k, t) = w1[k]; });
          }
        }
      }
      CSDO_XT(5);   // w pass
      CSDO_SYNC();
      CSDO_XT(6);   // wait for the tail product
      CSDO_PHASE(8);
#if defined(CSDO_ABL_XNOBWD)
      if (false) {
#else
      if (h_tail > 1) {
#endif
        int lev = 0;
        for (int h = 1; 2 * h < h_tail; h <<= 1) ++lev;
        for (int h = h_tail >> 1; h >= 2; h >>= 1, --lev) {
          CSDO_XLANES(t) {
            SolvRegs& V = CSDO_SS(t);
            const int tl = t < NtE ? t : NtE - 1;
            // the node's lane takes x of the left neighbour, its partner x of the right one - or, for a node without one, of
            // the left one again (second half of F_l, see the factorisation); lanes not at work read any valid slot
            const int n = t & ~1;
            int src = (t & 1) ? (((n + h) < Nt) ? n + h : n - h) : t - h;
            src = src < 0 ? 0 : (src >= NtE ? NtE - 1 : src);
            CSDO_FOR(k, 6, { V.v[k] = SH(vec, k, src); });
            if (V.fl & (XF_NR << lev)) CSDO_FOR(k, 3, { V.v[3 + k] = 0.0; });
            CSDO_FOR(r, 6, { V.o[r] = V.b[r]; });
            if constexpr (MODE == 2) {   // (two columns of the block at a time, see the forward sweep)
              CSDO_FOR(c2, 3, {
                double cw[12];
                CSDO_FOR(e, 12, { cw[e] = (c2 * 12 + e) < XER ? V.er[(c2 * 12 + e) < XER ? (c2 * 12 + e) : 0] : A2_LDS((c2 * 12 + e) >= XER ? (c2 * 12 + e) - XER : 0, tl); });
                CSDO_FOR(ch, 2, {
                  CSDO_FOR(r, 6, { V.o[r] = fma(-cw[ch * 6 + r], V.v[2 * c2 + ch], V.o[r]); });
                });
                CSDO_STAGE();
              });
            } else {
              double m2x[XFX > 0 ? XFX : 1];
              CSDO_FOR(k, XFX, { m2x[k] = A2_LDS(k, tl); });
              CSDO_FOR(c, 6, {
                CSDO_FOR(r, 6, { V.o[r] = fma(-A2(c * 6 + r), V.v[c], V.o[r]); });
              });
            }
          CSDO_XSTEP(t)
            double ub[6];
            CSDO_FOR(k, 6, { ub[k] = CSDO_XGET(CSDO_DPP_PAIR_SWAP, o, k); });
            if (V.fl & (XF_OWN << lev)) CSDO_FOR(k, 6, { SH(vec, k, t) = V.o[k] + ub[k]; });
          CSDO_XSTEP_LDS(t)
            (void)V;
          }
        }
        CSDO_ONCE_LOOP
        CSDO_XLANES(t) {   // level 1: x of the odd nodes, summed and stored by the even lane of the pair
          SolvRegs& V = CSDO_SS(t);
          if constexpr (MODE == 2) CSDO_FOR(k, 36, { V.el[k] = FE(k, t < NtE ? t : NtE - 1); });
          int src = (t & 1) ? (((t + 1) < Nt) ? t + 1 : t - 1) : t;
          src = src >= NtE ? NtE - 1 : src;
          CSDO_FOR(k, 6, { V.v[k] = SH(vec, k, src); });
          if (V.fl & XF_NR) CSDO_FOR(k, 3, { V.v[3 + k] = 0.0; });
          if (V.fl & XF_OWN) CSDO_FOR(k, 6, { V.b[k] = SH(pr, k, t); });   // even lane: w of the odd node; odd lanes start from their 0
          CSDO_FOR(r, 6, { V.o[r] = V.b[r]; });
          CSDO_FOR(c, 6, {
            CSDO_FOR(r, 6, { V.o[r] = fma(-V.el[c * 6 + r], V.v[c], V.o[r]); });
          });
        CSDO_XSTEP(t)
          double ub[6];
          CSDO_FOR(k, 6, { ub[k] = CSDO_XGET(CSDO_DPP_PAIR_ODD, o, k); });
          if (V.fl & XF_OWN) CSDO_FOR(k, 6, { SH(vec, k, t + 1) = V.o[k] + ub[k]; });
        }
        CSDO_ONCE_END
      }
      CSDO_XT(7);   // backward sweep
      CSDO_PHASE(14);
      before_last_barrier();
      CSDO_SYNC();
      CSDO_PHASE(8);
#undef A2
    };

    auto solve_lds = [&](auto) __attribute__((always_inline)) {
      CSDO_MARK("solve_begin");
      CSDO_PHASE(7);
#if defined(CSDO_ABL_NOSOLVE)
      CSDO_SYNC();
      return;
#endif
      for (int h = 1; h < h_tail; h <<= 1) {
        const int m2 = 2 * h - 1;
        CSDO_SLANES(t) {
          SolvRegs& V = CSDO_SS(t);
          if (h > 1 && (t & (h - 1)) == 0) {  // absorb the partials of the previous level
            const int hp = h >> 1;
            if (t >= hp) CSDO_FOR(k, 6, { V.b[k] -= SH(pr, k, t - hp); });
            if ((t + hp) < Nt) CSDO_FOR(k, 6, { V.b[k] -= SH(pl, k, t + hp); });
          }
#if defined(CSDO_ABL_NOLEVELWORK)
          if (false) {
#else
          if ((t & m2) == h) {
#endif
            CSDO_LVL_BEGIN();
            // three independent 6x6 products of the same b: pl = F_l' b, pr = F_r b (couplings in registers: published
            // first, the neighbours wait for them), then w = Sinv b with the pivot inverse fetched from LDS last, so that
            // its 21 doubles are not live beside the other products' accumulators.
            // fp64 FMAs need >= ~11 independent accumulation chains to issue back to back (measured: 6 chains run
            // at ~7 cycles per FMA, scripts/microbench.hip), so every product is split into two half-sums.
            const bool has_r = (t + h) < Nt;
            double bb[6];
            CSDO_FOR(k, 6, { bb[k] = V.b[k]; });
            {
              double pa[6] = {0, 0, 0, 0, 0, 0}, pb[6] = {0, 0, 0, 0, 0, 0};
              CSDO_FOR(r, 3, {
                CSDO_FOR(c, 6, {
                  pa[c] = fma(V.el[r * 6 + c], bb[r], pa[c]);
                  pb[c] = fma(V.el[(r + 3) * 6 + c], bb[r + 3], pb[c]);
                });
              });
              CSDO_FOR(c, 6, { SH(pl, c, t) = pa[c] + pb[c]; });
            }
            if (has_r) {
              double er_[36];
              CSDO_FOR(k, 36, { er_[k] = ER(k, t); });
              double qa[6] = {0, 0, 0, 0, 0, 0}, qb[6] = {0, 0, 0, 0, 0, 0};
              CSDO_FOR(k, 3, {
                CSDO_FOR(arow, 6, {
                  qa[arow] = fma(er_[arow * 6 + k], bb[k], qa[arow]);
                  qb[arow] = fma(er_[arow * 6 + k + 3], bb[k + 3], qb[arow]);
                });
              });
              CSDO_FOR(arow, 6, { SH(pr, arow, t) = qa[arow] + qb[arow]; });
            }
            // (w = Sinv b is only needed by the backward sweep: every eliminated node computes it in ONE pass behind the
            // last level, while the tail nodes gather - inside the levels it cost each of them ~400 cycles of critical path)
            CSDO_LVL_END(lvl_fwd);
          }
        }
        CSDO_PHASE(h == 1 ? 20 : 13);   // (diagnostic build: the first interval also holds the rhs assembly)
        CSDO_SYNC();
        CSDO_PHASE(7);
      }
      // ---- dense tail: tail nodes absorb the last level's partials and gather their rhs; tail lane r multiplies row r
      // of the explicit inverse with it and scatters the solution straight into the nodes' vec slots
      CSDO_SLANES(t) {
        SolvRegs& V = CSDO_SS(t);
        if ((t & (h_tail - 1)) == 0) {
          if (h_tail > 1) {
            const int hp = h_tail >> 1;
            if (t >= hp) CSDO_FOR(k, 6, { V.b[k] -= SH(pr, k, t - hp); });
            if ((t + hp) < Nt) CSDO_FOR(k, 6, { V.b[k] -= SH(pl, k, t + hp); });
          }
          const int kn = t / h_tail;
          CSDO_FOR(k, 6, { sh.tvec[6 * kn + k] = V.b[k]; });
        }
      }
      CSDO_PHASE(21);
      CSDO_SYNC();
      CSDO_PHASE(22);
#if defined(CSDO_LANE_MODE_DEVICE)
      // The product of the explicit tail inverse with the gathered rhs runs on the ROW waves, which have nothing else to do
      // between these two barriers: three lanes per row of the inverse, each with two of the row's six accumulation chains
      // (entry j of a row belongs to chain j mod 6, as in the one-lane-per-row form below: same operations in the same order,
      // same bits), joined by two shuffles.  One lane per row was 36 LDS round trips and FMAs deep: 1.5 k cycles of every iteration.
      // (Not in the 1024-thread class, residency mode 3: with 128 registers per lane the extra code costs its row waves more
      //  than the product takes on the solver wave - room50's long-horizon agents: 91.9 -> 88.0 ms.)
      if constexpr (ROLE == ROLE_ROW && MODE < 2) {
        const int x = CSDO_TID_HOT, wl = x & 63, r = (x >> 6) * 21 + wl / 3, p2 = 2 * (wl % 3);
        if (x < 128) {
          double a0 = 0.0, a1 = 0.0;
          if (wl < 63 && r < n_tail) {
            double tr0[6], tr1[6], tb0[6], tb1[6];
            CSDO_FOR(j, 6, {
              tr0[j] = TINV(6 * j + p2, r);
              tr1[j] = TINV(6 * j + p2 + 1, r);
              tb0[j] = sh.tvec[6 * j + p2];
              tb1[j] = sh.tvec[6 * j + p2 + 1];
            });
            CSDO_FOR(j, 6, {
              a0 = fma(tr0[j], tb0[j], a0);
              a1 = fma(tr1[j], tb1[j], a1);
            });
          }
          const double s01 = a0 + a1;                                   // lane 0 of the row: a0 + a1, lane 1: a2 + a3, lane 2: a4 + a5
          // (the neighbours' sums by DPP wave_shl:1 - a register move -, not by __shfl_down: that is a ds_bpermute, an LDS round trip
          //  each, two in a row on the critical path of every iteration)
          const double up1 = wave_next(s01), up2 = wave_next(up1);       // s01 of lane + 1, of lane + 2
          const double s0123 = s01 + up1;                               // lane 0: (a0 + a1) + (a2 + a3)
          const double tot = s0123 + up2;                               // lane 0: ... + (a4 + a5)
          if (wl < 63 && r < n_tail && p2 == 0) {
            const int kn = r / 6, i = r - 6 * kn;
            sh.vec[(kn * h_tail) * LD_vec + i] = tot;
          }
        }
      }
      if constexpr (ROLE == ROLE_BOTH || MODE >= 2)
#endif
      CSDO_TLANES_TOP(t) {
        {
          // columns >= n_tail of the inverse rows and of the gathered rhs are zero: no per-column test needed
          double a4[6] = {0, 0, 0, 0, 0, 0};
          if constexpr (BIGT) {
            for (int c0 = 0; c0 < ldt - 2; c0 += 6)
              CSDO_FOR(i6, 6, { a4[i6] = fma(TINV(c0 + i6, t), sh.tvec[c0 + i6], a4[i6]); });
          } else
          CSDO_FOR(q, 4, {     // quarters of the row: the solver lanes' registers hold their node's factor
            double tr[TAIL_N / 4], tb[TAIL_N / 4];
            CSDO_FOR(c, TAIL_N / 4, {
              tr[c] = TINV(q * (TAIL_N / 4) + c, t);
              tb[c] = sh.tvec[q * (TAIL_N / 4) + c];
            });
            CSDO_FOR(c, TAIL_N / 4, { a4[(q * (TAIL_N / 4) + c) % 6] = fma(tr[c], tb[c], a4[(q * (TAIL_N / 4) + c) % 6]); });
            CSDO_STAGE();
          });
          const int kn = t / 6, i = t - 6 * kn;
          sh.vec[(kn * h_tail) * LD_vec + i] = ((a4[0] + a4[1]) + (a4[2] + a4[3])) + (a4[4] + a4[5]);
        }
      }
      CSDO_SLANES(t) {   // beside the tail product (which runs on the workgroup's last wave): w of the eliminated nodes
        SolvRegs& V = CSDO_SS(t);
        if ((t & (h_tail - 1)) != 0) {
          // w = Sinv b of every eliminated node, three rows at a time: 15 of the 21 packed entries and 9 accumulation chains
          // are live at once (with all 21 + 12 chains beside the couplings the allocator spills a third of the factor)
          double bb[6], w6[6];
          CSDO_FOR(k, 6, { bb[k] = V.b[k]; });
          CSDO_FOR(half, 2, {
            double sv[6][6];
            CSDO_FOR(r3, 3, {
              CSDO_FOR(c, 6, { sv[r3][c] = SINV(sym(3 * half + r3, c), t); });
            });
            CSDO_FOR(r3, 3, {
              const double s01 = fma(sv[r3][1], bb[1], sv[r3][0] * bb[0]);
              const double s23 = fma(sv[r3][3], bb[3], sv[r3][2] * bb[2]);
              const double s45 = fma(sv[r3][5], bb[5], sv[r3][4] * bb[4]);
              w6[3 * half + r3] = (s01 + s23) + s45;
            });
            CSDO_STAGE();
          });
          CSDO_FOR(k, 6, { V.b[k] = w6[k]; });
        }
      }
      CSDO_SYNC();
      CSDO_PHASE(8);
      CSDO_SLANES(t) {
        SolvRegs& V = CSDO_SS(t);
        if ((t & (h_tail - 1)) == 0) CSDO_FOR(k, 6, { V.b[k] = SH(vec, k, t); });
      }
      for (int h = h_tail >> 1; h >= 1; h >>= 1) {
        const int m2 = 2 * h - 1;
        CSDO_SLANES(t) {
          SolvRegs& V = CSDO_SS(t);
#if defined(CSDO_ABL_NOLEVELWORK) || defined(CSDO_ABL_NOBWDWORK)
          if (false) {
#else
          if ((t & m2) == h) {
#endif
            CSDO_LVL_BEGIN();
            // x = w - F_l x_left - F_r' x_right: two independent half-sums (12 accumulation chains)
            const bool has_r = (t + h) < Nt;
            double xl[6], xr[6] = {0, 0, 0, 0, 0, 0};
            CSDO_FOR(k, 6, { xl[k] = SH(vec, k, t - h); });
            if (has_r) CSDO_FOR(k, 6, { xr[k] = SH(vec, k, t + h); });
            double ua[6], ub[6] = {0, 0, 0, 0, 0, 0};
            CSDO_FOR(k, 6, { ua[k] = V.b[k]; });
            if (has_r) {
              double er_[36];
              CSDO_FOR(k, 36, { er_[k] = ER(k, t); });
              CSDO_FOR(c, 6, {
                CSDO_FOR(r, 6, {
                  ua[r] = fma(-V.el[r * 6 + c], xl[c], ua[r]);
                  ub[r] = fma(-er_[c * 6 + r], xr[c], ub[r]);
                });
              });
            } else {
              CSDO_FOR(c, 3, {
                CSDO_FOR(r, 6, {
                  ua[r] = fma(-V.el[r * 6 + c], xl[c], ua[r]);
                  ub[r] = fma(-V.el[r * 6 + c + 3], xl[c + 3], ub[r]);
                });
              });
            }
            CSDO_FOR(k, 6, {
              const double xk = ua[k] + ub[k];
              V.b[k] = xk;
              SH(vec, k, t) = xk;
            });
            CSDO_LVL_END(lvl_bwd);
          }
        }
        CSDO_PHASE(14);
        CSDO_SYNC();
        CSDO_PHASE(8);
      }
    };

    // `before_last_barrier`: what the row waves do while they wait for the backward sweep (pair-split modes)
    auto solve = [&](auto before_last_barrier) __attribute__((always_inline)) {
      if constexpr (MODE != 3) solve_pair(before_last_barrier);
      else solve_lds(0);
    };

    // (one site for the factorisation, at the top of the block loop: every inlined copy of a large cold piece takes part in
    // the register allocation of the whole program - measured with the box code, dsqp_program.h)
    bool need_factor = true;

    // ============================================================== ADMM (osqp_solve, osqp.c)
    const int max_it = P.osqp_max_iter;
    const int chk = P.check_termination;
    int iter = 0;
    int qp_status = -10;  // OSQP_UNSOLVED
    double nrm[12];       // last update_info: see the residual block below
    CSDO_FOR(k, 12, { nrm[k] = 0.0; });

    const int K_planes = ad.n_planes;
    const bool rows_lds = (MODE == 0) && (uniform_i32(ad.rows_lds) != 0);
    // (re-typed: through the Shm field these were flat accesses - a 64-bit address per value in vector registers)
    double* const pco_lds = (MODE == 0) ? lds_ptr(sh.pco) : nullptr;
#define PC_L(k, p) sh.pc[(p) * 3 + (k)]
#define PC_G(k, p) sh.pcg[(p) * 3 + (k)]
#define PROW(f, p) sh.prow[(p) * LD_prow + (f)]   // f: 0..3 y, 4..7 z, 8 timestep
#define PCO(f, p) pco_lds[(f) * sh.n_pco_ld + (p)]
    // primal infeasibility certificate test (auxil.c is_primal_infeasible); collective.  Like update_info it runs right
    // after a block: coefficients come from the row-lane registers, delta_y / bounds / scalings from the workspace in
    // one batch of loads.
    auto primal_infeasible = [&](const double eps_pinf) __attribute__((always_inline)) -> bool {
      CSDO_MARK("pinf_begin");
      CSDO_LANES(t) {
        const unsigned act_ = (unsigned)WS(W_ACT, t);
        double dy[NROW], lo_[NROW], hi_[NROW], ee[NROW];
        CSDO_FOR(i, NROW, {
          dy[i] = WS(C_DY + i, t);
          lo_[i] = WS(W_LO + i, t);
          hi_[i] = WS(W_HI + i, t);
          ee[i] = WS(C_E + i, t);
        });
        double nmax = 0.0, acc = 0.0;
        CSDO_FOR(i, NROW, {
          if (act_ & (1u << i)) {
            double dyi = dy[i];
            if (hi_[i] > OSQP_INFTY * MIN_SCALING) {
              if (lo_[i] < -OSQP_INFTY * MIN_SCALING) dyi = 0.0;
              else dyi = osqp_min(dyi, 0.0);
            } else if (lo_[i] < -OSQP_INFTY * MIN_SCALING) {
              dyi = osqp_max(dyi, 0.0);
            }
            WS(C_DY + i, t) = dyi;
            nmax = dmax(nmax, fabs(ee[i] * dyi));
            acc += hi_[i] * osqp_max(dyi, 0.0) + lo_[i] * osqp_min(dyi, 0.0);
          }
        });
        // reference quirk: l = -infinity (a true IEEE inf, dsqp_solver.cc:1121-1123) times min(dy,0) = 0 is NaN, so
        // the certificate test is false for every agent that has inter-vehicle rows; IEEE arithmetic reproduces it
        const double ninf = -INFINITY;
        for (int r = 4 * tstart[t]; r < 4 * tstart[t + 1]; ++r) {  // l = -inf, u finite
          const double dy_r = ROW(r, R_DY), e_r = ROW(r, R_E), u_r = ROW(r, R_U);   // (all three in flight before the store)
          const double d = osqp_max(dy_r, 0.0);
          ROW(r, R_DY) = d;
          nmax = dmax(nmax, fabs(e_r * d));
          acc += u_r * osqp_max(d, 0.0) + ninf * osqp_min(d, 0.0);
        }
        const double part[2] = {nmax, acc};
        red_put<2>(sh, t, part);
      }
      // slot 0 is a max, slot 1 a sum: fold both ways over the same scratch
      double r_max[2], r_sum[2];
      red_fold<2, false>(sh, Nt, r_max);
      const double norm_dy = r_max[0];
      if (!(norm_dy > eps_pinf)) return false;
      red_fold<2, true>(sh, Nt, r_sum);
      if (!(r_sum[1] < -eps_pinf * norm_dy)) return false;
      // || Dinv A' dy ||
      CSDO_LANES(t) {
        const unsigned act_ = (unsigned)WS(W_ACT, t);
        CSDO_FOR(k, 4, { SH(carry, k, t) = (act_ & (1u << k)) ? WS(W_CN + k, t) * WS(C_DY + k, t) : 0.0; });
      }
      CSDO_SYNC();
      CSDO_LANES(t) {
        const unsigned act_ = (unsigned)WS(W_ACT, t);
        const int ncols_ = (t < Nm) ? 6 : 4;
        double dy[NROW];
        CSDO_FOR(i, NROW, { dy[i] = WS(C_DY + i, t); });
        double v[6] = {0, 0, 0, 0, 0, 0};
        if (t > 0) CSDO_FOR(k, 4, { v[k] = SH(carry, k, t - 1); });
        CSDO_FOR(i, NROW, {
          if (act_ & (1u << i)) {
            CSDO_FOR(s, 3, {
              if constexpr (row_col(i, s) >= 0) v[row_col(i, s)] = fma(WS(W_C + 3 * i + s, t), dy[i], v[row_col(i, s)]);
            });
          }
        });
        for (int r = 4 * tstart[t]; r < 4 * tstart[t + 1]; ++r) {
          const double d = ROW(r, R_DY);
          v[0] = fma(ROW(r, R_CA), d, v[0]);
          v[1] = fma(ROW(r, R_CB), d, v[1]);
          v[2] = fma(ROW(r, R_CY), d, v[2]);
        }
        double nmax = 0.0;
        CSDO_FOR(j, 6, {
          if (j < ncols_) nmax = dmax(nmax, fabs((1.0 / WS(C_D + j, t)) * v[j]));
        });
        const double part[1] = {nmax};
        red_put<1>(sh, t, part);
      }
      double r3[1];
      red_fold<1, false>(sh, Nt, r3);
      return r3[0] < eps_pinf * norm_dy;
    };

    // auxil.c check_termination on the residuals in nrm[]; returns true if a status was set
    auto check_termination = [&](const bool approximate) __attribute__((always_inline)) -> bool {
      double eps_abs = P.eps_abs, eps_rel = P.eps_rel, eps_pinf = P.eps_prim_inf;
      const double pri_res = nrm[0], dua_res = cinv * nrm[6];
      if (pri_res > OSQP_INFTY || dua_res > OSQP_INFTY) {
        qp_status = -7;
        return true;
      }
      if (approximate) {
        eps_abs *= 10;
        eps_rel *= 10;
        eps_pinf *= 10;
      }
      const double eps_prim = eps_abs + eps_rel * osqp_max(nrm[1], nrm[2]);
      bool prim_ok = false, prim_inf = false;
      if (pri_res < eps_prim) prim_ok = true;
      // With at least one inter-vehicle row the certificate sum contains (-inf) * 0 = NaN (see primal_infeasible), so the
      // test is false whatever the iterate: skip its three reductions.  (Its only side effect, the projection of
      // delta_y, is not observable: delta_y is rewritten by the next iteration.)
      else prim_inf = has_inter ? false : primal_infeasible(eps_pinf);
      double mx = osqp_max(0.0, nrm[7]);  // ||Dinv q|| = 0, then ||Dinv A'y||, then ||Dinv P x||
      mx = osqp_max(mx, nrm[8]);
      const double eps_dual = eps_abs + eps_rel * (mx * cinv);
      const bool dual_ok = dua_res < eps_dual;
      // dual infeasibility needs q'dx < 0; q = 0 in this problem family (:196-197), so it can never trigger
      if (prim_ok && dual_ok) {
        qp_status = approximate ? 2 : 1;
        return true;
      }
      if (prim_inf) {
        qp_status = approximate ? 3 : -3;
        return true;
      }
      return false;
    };

    // update_info (auxil.c): residuals of the current (x, z, y), unscaled for the termination test and scaled for
    // adapt_rho.  Runs once after every block, while the row lanes still hold the block's state (coefficients, x, y, z) in
    // registers; only the scalings come from the workspace.  The inter-vehicle rows' duals and slacks are read where the block
    // kept them (LDS for rows_lds agents), their coefficients from the block's LDS copy where it has one.
    auto update_info = [&]() __attribute__((always_inline)) {
      CSDO_MARK("info_begin");
      CSDO_PHASE(10);
      CSDO_LANES_HOT(t) {
        LaneState& S = CSDO_LS(t);
        CSDO_FOR(k, 5, { SH(carry2, k, t) = S.x[k]; });   // to t-1: x_{t+1} cols 0..3 and v_{t+1}
        // to t+1: A'y share of the kinematic rows (rows that do not exist have cn = y = 0)
        CSDO_FOR(k, 4, { SH(carry, k, t) = S.cn[k] * S.y[k]; });
        SH(carry, 4, t) = S.x[4];
        SH(carry, 5, t) = WS(W_P + 2, t);
      }
      CSDO_SYNC();
      CSDO_SUB_RESET();
      CSDO_LANES_HOT(t) {
        LaneState& S = CSDO_LS(t);
        const int ncols_ = (t < Nm) ? 6 : 4;
        double einv[NROW], dinv[6];
        CSDO_FOR(i, NROW, { einv[i] = WS(C_E + i, t); });
        CSDO_FOR(j, 6, { dinv[j] = WS(C_D + j, t); });
        const double pvv = WS(W_P + 0, t), pww = WS(W_P + 1, t), pvn = WS(W_P + 2, t);
        const int k0 = tstart[t], k1 = tstart[t + 1];
        double xn[4] = {0, 0, 0, 0};
        double vn = 0.0, vp = 0.0, pvn_left = 0.0;
        if (t < Nm) {
          CSDO_FOR(k, 4, { xn[k] = SH(carry2, k, t + 1); });
          vn = SH(carry2, 4, t + 1);
        }
        double Aty[6] = {0, 0, 0, 0, 0, 0};
        if (t > 0) {
          CSDO_FOR(k, 4, { Aty[k] = SH(carry, k, t - 1); });
          vp = SH(carry, 4, t - 1);
          pvn_left = SH(carry, 5, t - 1);
        }
        double p[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        CSDO_FOR(i, NROW, {   // (every row: the ones that do not exist at this t are all-zero with E = 1 and change no maximum)
          {
            double ax = 0.0;
            CSDO_FOR(s, 3, {
              if constexpr (row_col(i, s) >= 0) ax = fma(S.c[i][s], S.x[row_col(i, s)], ax);
            });
            if constexpr (i < 4) ax = fma(S.cn[i], xn[i], ax);
            const double zi = S.z[i], yi = S.y[i];
            const double ei = 1.0 / einv[i];
            const double res = ax - zi;
            p[0] = nmax(p[0], fabs(ei * res));
            p[1] = nmax(p[1], fabs(ei * zi));
            p[2] = nmax(p[2], fabs(ei * ax));
            p[3] = nmax(p[3], fabs(res));
            p[4] = nmax(p[4], fabs(zi));
            p[5] = nmax(p[5], fabs(ax));
            CSDO_FOR(s, 3, {
              if constexpr (row_col(i, s) >= 0) Aty[row_col(i, s)] = fma(S.c[i][s], yi, Aty[row_col(i, s)]);
            });
          }
        });
        for (int k = k0; k < k1; ++k) {
          double ca[4], cb[4], cy[4], zz[4], yy[4], ee[4];
          CSDO_FOR(q, 4, { ee[q] = ROW(4 * k + q, R_E); });
          if (rows_lds) {
            CSDO_FOR(q, 4, {
              yy[q] = PROW(q, k);
              zz[q] = PROW(4 + q, k);
            });
            const int kk = (k < sh.n_pco) ? k : 0;
            CSDO_FOR(q, 4, {
              ca[q] = PCO(q, kk);
              cb[q] = PCO(4 + q, kk);
              cy[q] = PCO(8 + q, kk);
            });
            if (k >= sh.n_pco) {
              CSDO_FOR(q, 4, {
                ca[q] = csdo_keep_f64(ROW(4 * k + q, R_CA));
                cb[q] = csdo_keep_f64(ROW(4 * k + q, R_CB));
                cy[q] = csdo_keep_f64(ROW(4 * k + q, R_CY));
              });
            }
          } else {
            CSDO_FOR(q, 4, {
              zz[q] = ROW(4 * k + q, R_Z);
              yy[q] = ROW(4 * k + q, R_Y);
              ca[q] = ROW(4 * k + q, R_CA);
              cb[q] = ROW(4 * k + q, R_CB);
              cy[q] = ROW(4 * k + q, R_CY);
            });
          }
          CSDO_FOR(q, 4, {
            const double ax = (ca[q] * S.x[0] + cb[q] * S.x[1]) + cy[q] * S.x[2];
            const double ei = 1.0 / ee[q];
            const double res = ax - zz[q];
            p[0] = nmax(p[0], fabs(ei * res));
            p[1] = nmax(p[1], fabs(ei * zz[q]));
            p[2] = nmax(p[2], fabs(ei * ax));
            p[3] = nmax(p[3], fabs(res));
            p[4] = nmax(p[4], fabs(zz[q]));
            p[5] = nmax(p[5], fabs(ax));
            Aty[0] = fma(ca[q], yy[q], Aty[0]);
            Aty[1] = fma(cb[q], yy[q], Aty[1]);
            Aty[2] = fma(cy[q], yy[q], Aty[2]);
          });
        }
        double Px[6] = {0, 0, 0, 0, 0, 0};
        if (t < Nm) {
          Px[4] = (pvv * S.x[4] + pvn * vn) + pvn_left * vp;
          Px[5] = pww * S.x[5];
        }
        CSDO_FOR(j, 6, {
          if (j < ncols_) {
            const double dj = 1.0 / dinv[j];
            const double dr = (0.0 + Px[j]) + Aty[j];
            p[6] = nmax(p[6], fabs(dj * dr));
            p[7] = nmax(p[7], fabs(dj * Aty[j]));
            p[8] = nmax(p[8], fabs(dj * Px[j]));
            p[9] = nmax(p[9], fabs(dr));
            p[10] = nmax(p[10], fabs(Aty[j]));
            p[11] = nmax(p[11], fabs(Px[j]));
          }
        });
        red_put<12>(sh, t, p);
      }
      CSDO_SUB(3);
      red_fold<12, false>(sh, Nt, nrm);
      CSDO_SUB(4);
    };

    // ---- the ADMM loop runs in blocks that end where osqp_solve would look at the iterate (termination check,
    // rho adaptation, iteration cap).  Inside a block only register state and LDS are touched (agents whose planes do
    // not fit keep the inter-vehicle rows' state in the L2-resident workspace).
    // Inter-vehicle rows are spread evenly over ALL solver threads (also those beyond Nt), plane by plane (thread l takes
    // planes l, l + nthr, ...; at most ceil(K / nthr) planes per thread whatever their timesteps).  `plane_pass`
    // optionally applies the z / y update with x_tilde of the plane's timestep (read from sh.vec) and leaves the
    // plane's share of A'(rho z - y) in pc[p][0..2]; the solver lane of that timestep adds its planes' shares to its rhs.
    // ROWS_LDS: duals, slacks, the plane's timestep and the shares live in LDS for the block (Shm::prow, Shm::pc); the
    // rows' coefficients and bounds (constant over a QP) are read from the workspace through the vector cache.
    auto plane_pass = [&](auto update_c, auto keep_c, auto lds_c, const int lane, const int nthr, const double rho_now) __attribute__((always_inline)) {
      constexpr bool UPDATE = decltype(update_c)::value;
      constexpr bool KEEP = decltype(keep_c)::value;
      constexpr bool ROWS_LDS = decltype(lds_c)::value;
      const double rinv = uniform_f64(1.0 / rho_now);
      for (int p = lane; p < K_planes; p += nthr) {
        double zz[4], yy[4], ca[4], cb[4], cy[4], uu[4], xt[3] = {0, 0, 0};
        CSDO_FOR(q, 4, {
          if constexpr (ROWS_LDS) {
            yy[q] = PROW(q, p);
            zz[q] = PROW(4 + q, p);
          } else {
            zz[q] = ROW(4 * p + q, R_Z);
            yy[q] = ROW(4 * p + q, R_Y);
          }
        });
        // The block's copy of the read-only coefficients (Shm::pco) covers the first n_pco planes.  Written as "LDS or workspace"
        // per plane, the compiler merges the two sources into ONE flat load per value through a selected 64-bit address: sixteen
        // addresses, hoisted out of the iteration loop, spilled, and reloaded from scratch in every iteration's plane pass.  So:
        // an unconditional LDS read (index clamped into the copy), overridden from the workspace for the planes beyond it.
        if constexpr (ROWS_LDS && UPDATE) {
          const int pp = (p < sh.n_pco) ? p : 0;
          CSDO_FOR(q, 4, {
            ca[q] = PCO(q, pp);
            cb[q] = PCO(4 + q, pp);
            cy[q] = PCO(8 + q, pp);
            uu[q] = PCO(12 + q, pp);
          });
          if (p >= sh.n_pco) {
            CSDO_FOR(q, 4, {
              ca[q] = csdo_keep_f64(ROW(4 * p + q, R_CA));   // (kept apart from the LDS reads above: see the comment)
              cb[q] = csdo_keep_f64(ROW(4 * p + q, R_CB));
              cy[q] = csdo_keep_f64(ROW(4 * p + q, R_CY));
              uu[q] = csdo_keep_f64(ROW(4 * p + q, R_U));
            });
          }
        } else {
          CSDO_FOR(q, 4, {
            ca[q] = ROW(4 * p + q, R_CA);
            cb[q] = ROW(4 * p + q, R_CB);
            cy[q] = ROW(4 * p + q, R_CY);
            if constexpr (UPDATE) uu[q] = ROW(4 * p + q, R_U);
          });
        }
        if constexpr (UPDATE) {
          int tp;
          if constexpr (ROWS_LDS) tp = (int)PROW(8, p);
          else tp = planes[p].t;
          CSDO_FOR(k, 3, { xt[k] = SH(vec, k, tp); });
        }
        double ic[3] = {0, 0, 0};
        CSDO_FOR(q, 4, {
          double zq = zz[q], yq = yy[q];
          if constexpr (UPDATE) {
            const double ztr = (ca[q] * xt[0] + cb[q] * xt[1]) + cy[q] * xt[2];
            const double zr = alpha * ztr + (1.0 - alpha) * zq;
            const double zn = hot_min(zr + rinv * yq, uu[q]);  // lower bound is -inf
            const double d = rho_now * (zr - zn);
            if constexpr (KEEP) ROW(4 * p + q, R_DY) = d;
            yq = yq + d;
            zq = zn;
            if constexpr (ROWS_LDS) {
              PROW(q, p) = yq;
              PROW(4 + q, p) = zq;
            } else {
              ROW(4 * p + q, R_Y) = yq;
              ROW(4 * p + q, R_Z) = zq;
            }
          }
          const double g = fma(rho_now, zq, -yq);
          ic[0] = fma(ca[q], g, ic[0]);
          ic[1] = fma(cb[q], g, ic[1]);
          ic[2] = fma(cy[q], g, ic[2]);
        });
        if constexpr (ROWS_LDS) CSDO_FOR(k, 3, { PC_L(k, p) = ic[k]; });
        else CSDO_FOR(k, 3, { PC_G(k, p) = ic[k]; });
      }
    };

    bool can_check = false;
    bool finished = false;
    bool first_block = true;
    iter = 0;
    while (!finished) {
      if (need_factor) {
        factor(rho);
        need_factor = false;
      }
      int stop = max_it;
      if (chk) stop = osqp_min_i(stop, (iter / chk + 1) * chk);
      if (P.adaptive_rho_interval) stop = osqp_min_i(stop, (iter / P.adaptive_rho_interval + 1) * P.adaptive_rho_interval);
      CSDO_PHASE(12);
      const double rho_eq = uniform_f64(RHO_EQ_OVER_RHO_INEQ * rho), rinv_in = uniform_f64(1.0 / rho),
                   rinv_eq = uniform_f64(1.0 / rho_eq);
      // The row lane's own share of the next rhs of the reduced system, sigma x + A'(rho z - y) over its home rows (q = 0),
      // and the kinematic rows' share for t+1: computed where x, z, y have just been updated (registers), so that the
      // solve can start right behind the update's barrier.
      auto publish_rhs = [&](LaneState& S, const int t) __attribute__((always_inline)) {
        double r6[6];
        CSDO_FOR(j, 6, { r6[j] = (j < S.ncols) ? sigma * S.x[j] : 0.0; });
        double kin[4] = {0, 0, 0, 0};
        CSDO_FOR(i, NROW, {   // every row, also those that do not exist at this t: their c, cn, y, z are zero (see the update)
          const double g = fma(rho_row<i>(S, rho, rho_eq), S.z[i], -S.y[i]);
          CSDO_FOR(s_, 3, {
            if constexpr (row_col(i, s_) >= 0) r6[row_col(i, s_)] = fma(S.c[i][s_], g, r6[row_col(i, s_)]);
          });
          if constexpr (i < 4) kin[i] = S.cn[i] * g;
        });
        CSDO_FOR(j, 6, { SH(rhs, j, t) = r6[j]; });
        CSDO_FOR(k, 4, { SH(carry, k, t) = kin[k]; });
      };
      CSDO_SUB_RESET();
      CSDO_LANES_HOT(t) {
        // The row lane's state - coefficients, duals, slacks, iterate, row classes - stays in its registers from the warm start
        // to the end of the QP (the row waves only wait at barriers during a factorisation, and the residual update reads the
        // registers): no write-back after a block, no reload in front of the next (it was some 90 dependent workspace loads per
        // lane and block).  Only the bounds, read once per iteration, are staged into LDS again: the factorisation's exchange
        // columns and the residual update's hand-over overwrite them (layout: Shm::lohi).
        LaneState& S = CSDO_LS(t);
        if constexpr (MODE < 2) {   // (loads first, then the stores: see the solver lanes' staging below)
          double lo_[13], hi_[9];
          CSDO_FOR(i, 13, { lo_[i] = WS(W_LO + i, t); });
          CSDO_FOR(i, 9, { hi_[i] = WS(W_HI + 7 + i, t); });
          CSDO_STAGE();
          CSDO_FOR(i, 13, { SH(lohi, i, t) = lo_[i]; });
          CSDO_FOR(i, 9, { SH(lohi, 13 + i, t) = hi_[i]; });
        }
        publish_rhs(S, t);
      }
      if constexpr (MODE == 3) {
        CSDO_SLANES(t) {  // load the solver-lane cache: F_l of the node in registers
          SolvRegs& V = CSDO_SS(t);
          CSDO_FOR(k, 36, { V.el[k] = FE(k, t); });
          V.ts0 = csdo_keep(tstart[t]);       // (kept in registers: re-reading them costs an L2 round trip per iteration)
          V.ts1 = csdo_keep(tstart[t + 1]);
        }
      } else {
        // pair-split solve: the LANE's two blocks (as the factorisation left them for it), 36 + ER_REG doubles in registers, the
        // rest of the second block and the node's pivot inverse in LDS; what the lane does at which level as bit masks
        CSDO_XLANES(t) {
          SolvRegs& V = CSDO_SS(t);
          const int tl = t < NtE ? t : NtE - 1;   // (threads beyond the horizon take part in the wave's moves: any valid address)
          if constexpr (MODE != 2) CSDO_FOR(k, 36, { V.el[k] = FE(k, tl); });   // (mode 2 fetches it where it is used)
          CSDO_FOR(k, XER, { V.er[k] = FE(72 + k, tl); });
          {
            if (t < NtE) {
              // (all loads first, then the LDS stores: written element by element - load, store, load ... - every load was waited
              //  for before the next was issued: 27 trips to the workspace in a row at the head of every block)
              double stg[XFX > 0 ? XFX : 1], sinv_[21];
              CSDO_FOR(k, XFX, { stg[k] = FE(72 + XER + k, t); });
              if constexpr (MODE == 0) CSDO_FOR(k, 21, { sinv_[k] = WS(W_SINV + k, t); });
              CSDO_STAGE();
              CSDO_FOR(k, XFX, { A2_LDS(k, t) = stg[k]; });
              if constexpr (MODE == 0) CSDO_FOR(k, 21, { SH(fx, FX_ER + k, t) = sinv_[k]; });
              (void)sinv_;
            }
          }
          V.ts0 = V.ts1 = 0;
          if (t < Nt) {
            V.ts0 = csdo_keep(tstart[t]);       // (kept in registers: re-reading them costs an L2 round trip per iteration)
            V.ts1 = csdo_keep(tstart[t + 1]);
            // (in LDS - carry's two spare doubles - instead of lane state the rhs assembly loses its two scratch reloads and the step
            //  gets SLOWER, map100 57.99 -> 58.63 ms: the reloads were hidden, the LDS reads are not; round 5)
          }
          unsigned fl = 0;
          if (t < NtE) {
            const bool odd = (t & 1) != 0;
            int lev = 0;
            for (int h = 1; h < h_tail; h <<= 1, ++lev) {
              const int m2 = 2 * h - 1;
              if (!odd && t < Nt && (t & m2) == 0) {   // takes the level's partials: the right one always, the left one unless it
                if ((t & 63) != 0) fl |= XF_ABS << lev;   // comes from the wave in front (that wave hands the sum of them over, below)
                if ((t + h) < Nt) fl |= XF_ABSR << lev;
              }
              if (lev == 0) {
                // in-wave partials of level 1 travel by DPP; LDS only holds what the next wave's / this wave's first node takes later
                if (odd && (t & 63) == 63 && (t + 1) < Nt) fl |= XF_WR;
                if (!odd && (t + 1) < Nt) {
                  fl |= XF_OWN;
                  if ((t + 2) >= Nt) fl |= XF_NR;
                }
              } else {
                const int n = t & ~1;            // the node this pair works for at its level
                if ((n & m2) == h && n < Nt) {
                  if (!odd) {
                    fl |= (XF_WR | XF_OWN) << lev;
                    if ((n + h) >= Nt) fl |= XF_NR << lev;
                  } else if ((n + h) < Nt) {
                    fl |= XF_WR << lev;
                  }
                }
              }
            }
          }
          V.fl = fl;
          if constexpr (REFINE == 2) CSDO_FOR(j, 6, { V.rl[j] = 0.0; });
        }
      }
      if (rows_lds) {   // the inter-vehicle rows' duals, slacks and timesteps live in LDS for the whole QP (nothing else uses
        CSDO_STHREADS(l, nthr) {   // that part of it): staged in front of the QP's first block only
          for (int p = l; p < (first_block ? K_planes : 0); p += nthr) {
            CSDO_FOR(q, 4, {
              PROW(q, p) = ROW(4 * p + q, R_Y);
              PROW(4 + q, p) = ROW(4 * p + q, R_Z);
            });
            PROW(8, p) = (double)planes[p].t;
          }
          for (int p = l; p < (first_block ? sh.n_pco : 0); p += nthr) {
            CSDO_FOR(q, 4, {
              PCO(q, p) = ROW(4 * p + q, R_CA);
              PCO(4 + q, p) = ROW(4 * p + q, R_CB);
              PCO(8 + q, p) = ROW(4 * p + q, R_CY);
              PCO(12 + q, p) = ROW(4 * p + q, R_U);
            });
          }
          plane_pass(std::false_type{}, std::false_type{}, std::true_type{}, l, nthr, rho);
        }
      } else {
        CSDO_STHREADS(l, nthr) { plane_pass(std::false_type{}, std::false_type{}, std::false_type{}, l, nthr, rho); }
      }
      CSDO_SYNC();
      CSDO_SUB(5);
      CSDO_XT_RESET();
      auto iteration = [&](auto keep_c) __attribute__((always_inline)) {
        constexpr bool keep_dy = decltype(keep_c)::value;   // only the last iteration of a block records delta_y
        // ---- rhs of the reduced system: the solver lane adds the kinematic share of t-1 and its planes' shares to what
        // its row lane left in sh.rhs (previous update / block load); no row-lane phase, no barrier in between
        CSDO_MARK("rhs");
        CSDO_SLANES(t) {   // (the opaque lane index: with the plain one the LDS addresses of rhs and carry are hoisted out of the block's loop, spilled, and reloaded from scratch here - three waits in a row at the head of every iteration)
          SolvRegs& V = CSDO_SS(t);
          double r6[6];
          CSDO_FOR(j, 6, { r6[j] = SH(rhs, j, t); });
          if (t > 0) {   // (both halves in flight before the first is used)
            double cr[4];
            CSDO_FOR(k, 4, { cr[k] = SH(carry, k, t - 1); });
            CSDO_STAGE();
            CSDO_FOR(k, 4, { r6[k] += cr[k]; });
          }
          // the planes' shares, three planes per round trip (added one by one in the planes' order, as before: a timestep in a
          // cluster of vehicles has several planes, and one LDS - or workspace - round trip per plane was the wave's critical path)
          auto add_planes = [&](auto lds_c) __attribute__((always_inline)) {
            constexpr bool L = decltype(lds_c)::value;
            const int pe = V.ts1;
            for (int p = V.ts0; p < pe; p += 3) {
              double v[3][3];
              CSDO_FOR(q, 3, {
                const int pq = (p + q < pe) ? p + q : p;
                CSDO_FOR(k, 3, { v[q][k] = L ? PC_L(k, pq) : PC_G(k, pq); });
              });
              CSDO_FOR(q, 3, {
                const bool ok = (p + q) < pe;
                CSDO_FOR(k, 3, {
                  const double s_ = r6[k] + v[q][k];
                  r6[k] = ok ? s_ : r6[k];
                });
              });
            }
          };
          if (rows_lds) add_planes(std::true_type{});
          else add_planes(std::false_type{});
          CSDO_FOR(j, 6, { V.b[j] = r6[j]; });
          if constexpr (REFINE == 2) {   // lagged: what the previous iteration's solve missed rides on this rhs (zero at a block's start)
            CSDO_FOR(j, 6, { r6[j] += V.rl[j]; });
            CSDO_FOR(j, 6, { V.b[j] = r6[j]; });
          }
          if constexpr (REFINE != 0) CSDO_FOR(j, 6, { V.b0[j] = r6[j]; });
        }
        // (modes 2, 3: the row lanes stream their rows' coefficients and bounds from the workspace, see the update; on the device the
        //  first two groups of rows are fetched while the row waves wait for the backward sweep)
#define CSDO_ROWGROUP(name) double name##c[3][3], name##n[3], name##l[3], name##h[3]
        CSDO_ROWGROUP(ga); CSDO_ROWGROUP(gb); CSDO_ROWGROUP(gd);
#undef CSDO_ROWGROUP
        auto load_rows = [&](const int t, auto i0_c, auto n_c, double (&gc)[3][3], double (&gcn)[3], double (&glo)[3], double (&ghi)[3]) __attribute__((always_inline)) {
          constexpr int i0 = decltype(i0_c)::value, n = decltype(n_c)::value;
          CSDO_FOR(r, n, {
            constexpr int i = i0 + r;
            CSDO_FOR(s_, 3, {
              if constexpr (row_col(i, s_) >= 0) gc[r][s_] = WS(W_C + 3 * i + s_, t);
              else gc[r][s_] = 0.0;
            });
            if constexpr (i < 4) gcn[r] = WS(W_CN + i, t);
            else gcn[r] = 0.0;
            if constexpr (i < 13) glo[r] = WS(W_LO + i, t);
            if constexpr (i >= 7) ghi[r] = WS(W_HI + i, t);
            if constexpr (i < 7) ghi[r] = glo[r];
            if constexpr (i >= 13) glo[r] = -ghi[r];
          });
        };
#if defined(CSDO_LANE_MODE_DEVICE)
        constexpr bool rows_prefetched = (MODE == 2) && (ROLE == ROLE_ROW);
#else
        constexpr bool rows_prefetched = false;
#endif
        solve([&]() __attribute__((always_inline)) {
          if constexpr (rows_prefetched) {
            CSDO_LANES_HOT(t) {
              load_rows(t, std::integral_constant<int, 0>{}, std::integral_constant<int, 2>{}, gac, gan, gal, gah);
              load_rows(t, std::integral_constant<int, 2>{}, std::integral_constant<int, 2>{}, gbc, gbn, gbl, gbh);
            }
          }
        });
        if constexpr (REFINE != 0) {
          // ---- refinement (csdo_qp_parm::solve_refinement).  The reduced system H x = b, H = P + sigma I + A' R A, has the condition
          // of A' R A: a backward-stable solve of it - block cyclic reduction behaves like a Cholesky factorisation here - returns
          // x~ with |error| ~ cond(H) eps |x|, some fifty times the error of OSQP's LDL' of the quasi-definite KKT matrix
          // (scripts/solve_accuracy.py: 2e-12 .. 2e-11 against 3e-14 .. 3e-13 on the agent QPs of the pin kit), and over an SQP chain
          // that puts 1.3 - 1.5 times as many agents beyond 1e-4 of the exact-arithmetic path as a double-precision OSQP
          // (scripts/chain_parity.py).  One step of iterative refinement closes the gap IF the residual is that of
          // the KKT system, i.e. formed through A - r = b - (P + sigma) x~ - A' (R (A x~)), the large rho-weighted terms entering as
          // products of small numbers - and not through the formed blocks of H (measured: no gain).  Then x~ += H^-1 r with the same
          // factor: 3e-14 .. 2e-13.  Cost: a second solve and a pass over the rows per iteration, about twice the time.
          CSDO_LANES_HOT(t) {   // the home rows' share of H x~ (own columns -> rhs, column i of t + 1 -> carry; the arrays' contents
            LaneState& S = CSDO_LS(t);   // went into b, which the solver lane kept)
            double xt[6], xn[4] = {0, 0, 0, 0};
            CSDO_FOR(k, 6, { xt[k] = SH(vec, k, t); });
            if (t < Nm) CSDO_FOR(k, 4, { xn[k] = SH(vec, k, t + 1); });
            const double vnext = (t < Nm) ? SH(vec, 4, t + 1) : 0.0;
            const double pvv = WS(W_P + 0, t), pww = WS(W_P + 1, t), pvn = WS(W_P + 2, t);
            double r6[6], kin[4] = {0, 0, 0, 0};
            CSDO_FOR(j, 6, { r6[j] = (j < S.ncols) ? sigma * xt[j] : xt[j]; });   // (the padding of the last node: identity)
            r6[4] = fma(pvv, xt[4], r6[4]);     // P: the first-difference Laplacian on v (its coupling to t - 1: the solver lane below) ...
            r6[4] = fma(pvn, vnext, r6[4]);
            r6[5] = fma(pww, xt[5], r6[5]);     // ... and the identity on w
            CSDO_FOR(i, NROW, {
              double ci[3] = {0, 0, 0}, cni = 0.0;
              if constexpr (MODE < 2) {
                CSDO_FOR(s_, 3, { ci[s_] = S.c[i][s_]; });
                if constexpr (i < 4) cni = S.cn[i];
              } else {   // (the long-horizon classes stream the coefficients: from the workspace here)
                CSDO_FOR(s_, 3, {
                  if constexpr (row_col(i, s_) >= 0) ci[s_] = WS(W_C + 3 * i + s_, t);
                });
                if constexpr (i < 4) cni = WS(W_CN + i, t);
              }
              double zt = 0.0;
              CSDO_FOR(s_, 3, {
                if constexpr (row_col(i, s_) >= 0) zt = fma(ci[s_], xt[row_col(i, s_)], zt);
              });
              if constexpr (i < 4) zt = fma(cni, xn[i], zt);
              const double g = rho_row<i>(S, rho, rho_eq) * zt;
              CSDO_FOR(s_, 3, {
                if constexpr (row_col(i, s_) >= 0) r6[row_col(i, s_)] = fma(ci[s_], g, r6[row_col(i, s_)]);
              });
              if constexpr (i < 4) kin[i] = cni * g;
            });
            CSDO_FOR(j, 6, { SH(rhs, j, t) = r6[j]; });
            CSDO_FOR(k, 4, { SH(carry, k, t) = kin[k]; });
          }
          CSDO_SYNC();
          CSDO_SLANES(t) {   // r = b - H x~: the rows' shares, the hand-over from t - 1 and this timestep's inter-vehicle rows
            SolvRegs& V = CSDO_SS(t);
            double h6[6], x3[3];
            CSDO_FOR(j, 6, { h6[j] = SH(rhs, j, t); });
            CSDO_FOR(k, 3, { x3[k] = SH(vec, k, t); });
            if constexpr (REFINE == 1) CSDO_FOR(k, 6, { V.x0[k] = SH(vec, k, t); });
            if (t > 0) {
              CSDO_FOR(k, 4, { h6[k] += SH(carry, k, t - 1); });
              h6[4] = fma(WS(W_P + 2, t - 1), SH(vec, 4, t - 1), h6[4]);
            }
            for (int p = V.ts0; p < V.ts1; ++p) {
              CSDO_FOR(q, 4, {
                const double a = ROW(4 * p + q, R_CA), bb = ROW(4 * p + q, R_CB), cy = ROW(4 * p + q, R_CY);
                const double g = rho * ((a * x3[0] + bb * x3[1]) + cy * x3[2]);
                h6[0] = fma(a, g, h6[0]);
                h6[1] = fma(bb, g, h6[1]);
                h6[2] = fma(cy, g, h6[2]);
              });
            }
            if constexpr (REFINE == 1) CSDO_FOR(j, 6, { V.b[j] = V.b0[j] - h6[j]; });
            // LAGGED (solve_refinement = 2): no second solve - the residual joins the NEXT iteration's rhs, so that x~_{k+1} carries the
            // correction x~_k missed.  Between projections the ADMM state is linear in the x~'s: what one iteration lacks the next one
            // supplies, and the missing corrections telescope to the last one instead of adding up.  Measured against the arbiter
            // (scripts/chain_parity.py `product_lagged`): the distance of a double-precision OSQP, at 1.3 - 1.4 x the time instead of 1.8 x.
            else CSDO_FOR(j, 6, { V.rl[j] = V.b0[j] - h6[j]; });
          }
          CSDO_SYNC();
          if constexpr (REFINE == 1) {
            solve([]() __attribute__((always_inline)) {});
            CSDO_SLANES(t) {
              SolvRegs& V = CSDO_SS(t);
              double dx[6];
              CSDO_FOR(k, 6, { dx[k] = SH(vec, k, t); });
              CSDO_FOR(k, 6, { SH(vec, k, t) = V.x0[k] + dx[k]; });
            }
            CSDO_SYNC();
          }
        }
        CSDO_PHASE(9);
        // ---- x, z, y updates (update_x / update_z / update_y); delta_y is only consumed by the termination test
        CSDO_MARK("update");
#if !defined(CSDO_ABL_NOPLANES)
        if (rows_lds) {   // inter-vehicle rows, concurrently with the row lanes below
          CSDO_STHREADS_HOT(l, nthr) { plane_pass(std::true_type{}, keep_c, std::true_type{}, l, nthr, rho); }
        } else {
          CSDO_STHREADS_HOT(l, nthr) { plane_pass(std::true_type{}, keep_c, std::false_type{}, l, nthr, rho); }
        }
#endif
        CSDO_LANES_HOT(t) {
          LaneState& S = CSDO_LS(t);
          double xt[6], xn[4] = {0, 0, 0, 0};
          CSDO_FOR(k, 6, { xt[k] = SH(vec, k, t); });
          if (t < Nm) CSDO_FOR(k, 4, { xn[k] = SH(vec, k, t + 1); });
          // x first: the row loop below also accumulates the lane's share of the NEXT rhs, sigma x + A'(rho z - y) (the same
          // sums in the same order as publish_rhs), so that every coefficient is touched once per iteration
          double r6[6], kin[4] = {0, 0, 0, 0};
          CSDO_FOR(j, 6, {
            if (j < S.ncols) S.x[j] = alpha * xt[j] + (1.0 - alpha) * S.x[j];
            r6[j] = (j < S.ncols) ? sigma * S.x[j] : 0.0;
          });
          // No test for rows that do not exist at this t (kinematic and control rows at the last timestep, start / goal rows in
          // the interior): their coefficients, bounds, duals and slacks are all zero (assemble_home_rows, warm start) and stay
          // zero through the formulas below, and straight-line code lets the rows' dependent chains overlap (16 basic blocks
          // with an exec-mask test each could not).
          auto do_row = [&](auto i_c, const double c0, const double c1, const double c2, const double cni, const double lo_i,
                            const double hi_i) __attribute__((always_inline)) {
            constexpr int i = decltype(i_c)::value;
            const double ci[3] = {c0, c1, c2};
            double zt = 0.0;
            CSDO_FOR(s_, 3, {
              if constexpr (row_col(i, s_) >= 0) zt = fma(ci[s_], xt[row_col(i, s_)], zt);
            });
            if constexpr (i < 4) zt = fma(cni, xn[i], zt);
            const double rh = rho_row<i>(S, rho, rho_eq);
            const double rinv = rho_row<i>(S, rinv_in, rinv_eq);   // = 1.0 / rh (rho_inv_vec of OSQP)
            const double zr = alpha * zt + (1.0 - alpha) * S.z[i];
            const double zn = hot_min(hot_max(zr + rinv * S.y[i], lo_i), hi_i);
            const double d = rh * (zr - zn);
            if constexpr (keep_dy) WS(C_DY + i, t) = d;
            S.y[i] += d;
            S.z[i] = zn;
            const double g = fma(rh, zn, -S.y[i]);
            CSDO_FOR(s_, 3, {
              if constexpr (row_col(i, s_) >= 0) r6[row_col(i, s_)] = fma(ci[s_], g, r6[row_col(i, s_)]);
            });
            if constexpr (i < 4) kin[i] = cni * g;
          };
          if constexpr (MODE < 2) {
            CSDO_FOR(i, NROW, {
              double lo_i, hi_i;   // bounds in the packed order of Shm::lohi
              if constexpr (i < 7) lo_i = hi_i = SH(lohi, i, t);
              if constexpr (i >= 7 && i < 13) {
                lo_i = SH(lohi, i, t);
                hi_i = SH(lohi, i + 6, t);
              }
              if constexpr (i >= 13) {
                hi_i = SH(lohi, i + 6, t);
                lo_i = -hi_i;
              }
              do_row(i_c, S.c[i][0], S.c[i][1], S.c[i][2], (i < 4) ? S.cn[i < 4 ? i : 0] : 0.0, lo_i, hi_i);
            });
          } else {
            // Long horizons (168 or 128 registers per lane): duals, slacks and the iterate stay in registers, the rows' coefficients
            // and bounds - read-only during a QP, 53 doubles - are STREAMED from the workspace in seven groups of rows, the loads of
            // a group issued a group ahead of its use.  (With the coefficients among the lane's registers the allocator spilled a
            // third of the row state and reloaded it where it was used: 65 scratch loads per iteration, each waited for on its
            // own - 21 k of the 55 k cycles of an iteration of this class.)
            auto use_rows = [&](auto i0_c, auto n_c, const double (&gc)[3][3], const double (&gcn)[3], const double (&glo)[3], const double (&ghi)[3]) __attribute__((always_inline)) {
              constexpr int i0 = decltype(i0_c)::value, n = decltype(n_c)::value;
              CSDO_FOR(r, n, { do_row(std::integral_constant<int, i0 + r>{}, gc[r][0], gc[r][1], gc[r][2], gcn[r], glo[r], ghi[r]); });
            };
#define CSDO_LOAD_ROWS(name, I0_, N_) load_rows(t, std::integral_constant<int, I0_>{}, std::integral_constant<int, N_>{}, name##c, name##n, name##l, name##h); CSDO_STAGE()
#define CSDO_USE_ROWS(name, I0_, N_) use_rows(std::integral_constant<int, I0_>{}, std::integral_constant<int, N_>{}, name##c, name##n, name##l, name##h); CSDO_STAGE()
            if (!rows_prefetched) {
              CSDO_LOAD_ROWS(ga, 0, 2);
              CSDO_LOAD_ROWS(gb, 2, 2);
            }
            CSDO_USE_ROWS(ga, 0, 2);
            CSDO_LOAD_ROWS(gd, 4, 3);
            CSDO_USE_ROWS(gb, 2, 2);
            CSDO_LOAD_ROWS(ga, 7, 2);
            CSDO_USE_ROWS(gd, 4, 3);
            CSDO_LOAD_ROWS(gb, 9, 2);
            CSDO_USE_ROWS(ga, 7, 2);
            CSDO_LOAD_ROWS(gd, 11, 2);
            CSDO_USE_ROWS(gb, 9, 2);
            CSDO_LOAD_ROWS(ga, 13, 3);
            CSDO_USE_ROWS(gd, 11, 2);
            CSDO_USE_ROWS(ga, 13, 3);
#undef CSDO_LOAD_ROWS
#undef CSDO_USE_ROWS
          }
          CSDO_FOR(j, 6, { SH(rhs, j, t) = r6[j]; });
          CSDO_FOR(k, 4, { SH(carry, k, t) = kin[k]; });
        }
        CSDO_SYNC();
        CSDO_XT_RESET();
      };
      first_block = false;
      while (iter < stop - 1) {
        ++iter;
        iteration(std::false_type{});
      }
      ++iter;
      iteration(std::true_type{});
      update_info();
      CSDO_PHASE(12);
      CSDO_SYNC();

      // Every block ends where osqp_solve looks at the iterate (a termination check, a rho adaptation or the iteration cap), so
      // the residuals are formed after EVERY block, in ONE place: the twelve norms are then defined right here in every trip of
      // the loop and do not stay live across the iterations (24 scalar registers), and the large routine is inlined once.
      can_check = chk && (iter % chk == 0);
      if (can_check) {
#if !defined(CSDO_ABL_NOCHECK)
#if defined(CSDO_ABL_FIXED)
        check_termination(false);
        qp_status = -10;
#else
        if (check_termination(false)) break;
#endif
#endif
      }
      if (iter >= max_it) {
        finished = true;
      } else if (P.adaptive_rho_interval && (iter % P.adaptive_rho_interval == 0)) {
        // compute_rho_estimate (auxil.c), scaled residuals
        double pri = nrm[3], dua = nrm[9];
        const double pri_n = osqp_max(nrm[4], nrm[5]);
        pri /= (pri_n + 1e-10);
        double dua_n = osqp_max(0.0, nrm[10]);
        dua_n = osqp_max(dua_n, nrm[11]);
        dua /= (dua_n + 1e-10);
        double est = rho * sqrt(pri / (dua + 1e-10));
        est = osqp_min(osqp_max(est, RHO_MIN), RHO_MAX);
        if (est > rho * P.adaptive_rho_tolerance || est < rho / P.adaptive_rho_tolerance) {
          rho = uniform_f64(osqp_min(osqp_max(est, RHO_MIN), RHO_MAX));
          need_factor = true;
        }
      }
    }
    if (finished && qp_status == -10) {
      if (!can_check) check_termination(false);
    }
    if (qp_status == -10) {
      if (!check_termination(true)) qp_status = -2;  // OSQP_MAX_ITER_REACHED
    }
    status = qp_status;
    admm_total += iter;

  CSDO_PHASE(11);
    // ============================================================== SQP bookkeeping (calcIndividualSQP :223-253)
    CSDO_MARK("bookkeeping");
    CSDO_LANES(t) {
      LaneState& S = CSDO_LS(t);
      double acc = 0.0;
      const bool keep_prev = (status > 2 || status < -2);  // :515-524
      CSDO_FOR(j, 6, {
        if (j < S.ncols) {
          const double s0 = CD(C_SOL0 + j, t);
          const double sn = keep_prev ? s0 : CD(C_D + j, t) * S.x[j];
          CD(C_SOL + j, t) = sn;
          const double d = sn - s0;
          acc = fma(d, d, acc);
        } else {
          CD(C_SOL + j, t) = 0.0;
        }
      });
      const double part[1] = {acc};
      red_put<1>(sh, t, part);
    }
    {
      double r[1];
      red_fold<1, true>(sh, Nt, r);
      delta = r[0];
    }
    it++;

    bool feasible = false;
    if (it > P.max_iter / 2) {  // isFeasible :292-420
      CSDO_LANES(t) {
        CSDO_FOR(k, 4, { SH(carry2, k, t) = CD(C_SOL + k, t); });
      }
      CSDO_SYNC();
      CSDO_LANES(t) {
        double p[6] = {0, 0, 0, 0, 0, 0};
        const double x = CD(C_SOL + 0, t), y = CD(C_SOL + 1, t), yaw = CD(C_SOL + 2, t), stv = CD(C_SOL + 3, t),
                     v = CD(C_SOL + 4, t), w = CD(C_SOL + 5, t);
        const SinCos scw_ = sincos_of(yaw);
        const double syw = scw_.s, cyw = scw_.c;
        if (t < Nm) {
          const double r1 = x + v * cyw * P.dt - SH(carry2, 0, t + 1);
          const double r2 = y + v * syw * P.dt - SH(carry2, 1, t + 1);
          const double r3 = yaw + v * tan_of(stv) / P.WB * P.dt - SH(carry2, 2, t + 1);
          const double r4 = stv + w * P.dt - SH(carry2, 3, t + 1);
          p[0] = r1 * r1;
          p[1] = r2 * r2;
          p[2] = r3 * r3;
          p[3] = r4 * r4;
        }
        const double Y[4] = {x + P.f2x * cyw, y + P.f2x * syw, x + P.r2x * cyw, y + P.r2x * syw};
        double ecor = 0.0;
        CSDO_FOR(k, 4, {
          const double lbk = CD(C_CLB + k, t), ubk = CD(C_CUB + k, t);
          if (!(lbk <= Y[k])) ecor = dmax(ecor, lbk - Y[k]);
          if (!(Y[k] <= ubk)) ecor = dmax(ecor, Y[k] - ubk);
        });
        double eint = 0.0;
        for (int k = tstart[t]; k < tstart[t + 1]; ++k) {
          const PlaneDev& pl = planes[k];
          CSDO_FOR(r, 4, {
            const double px = (r < 2) ? Y[0] : Y[2], py = (r < 2) ? Y[1] : Y[3];
            const double res = ((0.0 + px * pl.c[3 * r]) + py * pl.c[3 * r + 1]) + pl.c[3 * r + 2];
            if (res > 0) eint = dmax(eint, res);
          });
        }
        p[4] = ecor;
        p[5] = eint;
        red_put<6>(sh, t, p);
      }
      double rs[4], rm[6];
      // slots 0..3 are sums, 4..5 maxima: fold both ways and pick
      {
        double all_sum[6];
        red_fold<6, true>(sh, Nt, all_sum);
        CSDO_FOR(k, 4, { rs[k] = all_sum[k]; });
      }
      // the scratch still holds the partials (red_fold does not modify them)
      red_fold<6, false>(sh, Nt, rm);
      const double err_kin = (((rs[0] + rs[1]) + rs[2]) + rs[3]) / (double)Nt;
      const double err_cor = rm[4];
      const double err_int = has_inter ? rm[5] : 0.0;
      feasible = (err_kin < 1e-2) && (err_int < 1e-1) && (err_cor < 1e-1);
    }
#if !defined(CSDO_ABL_FIXED)
    if (feasible) break;
#endif

    CSDO_PHASE(1);
    CSDO_SLANES(t) {  // updateCorridor :818-872 (double-precision disc centres): rear disc on the solver lane
      if (!P.fixed_corridor) {
        const double px = CD(C_SOL + 0, t), py = CD(C_SOL + 1, t), pyaw = CD(C_SOL + 2, t);
        const SinCos scp_ = sincos_of(pyaw);
        const double spy = scp_.s, cpy = scp_.c;
        const double xr = px + P.r2x * cpy, yr = py + P.r2x * spy;
        BoxD br;
        box_at(xr, yr, sh.obs, n_obs, dimx, dimy, rv, br, box_cache);
        CD(C_CLB + 2, t) = br.x_min; CD(C_CLB + 3, t) = br.y_min;
        CD(C_CUB + 2, t) = br.x_max; CD(C_CUB + 3, t) = br.y_max;
      }
    }
    CSDO_LANES(t) {  // solution0 = solution; front disc on the row lane
      if (!P.fixed_corridor) {
        const double px = CD(C_SOL + 0, t), py = CD(C_SOL + 1, t), pyaw = CD(C_SOL + 2, t);
        const SinCos scp_ = sincos_of(pyaw);
        const double spy = scp_.s, cpy = scp_.c;
        const double xf = px + P.f2x * cpy, yf = py + P.f2x * spy;
        BoxD bf;
        box_at(xf, yf, sh.obs, n_obs, dimx, dimy, rv, bf, box_cache);
        CD(C_CLB + 0, t) = bf.x_min; CD(C_CLB + 1, t) = bf.y_min;
        CD(C_CUB + 0, t) = bf.x_max; CD(C_CUB + 1, t) = bf.y_max;
      }
      CSDO_FOR(k, 6, { CD(C_SOL0 + k, t) = CD(C_SOL + k, t); });
    }
    CSDO_SYNC();
  }

  // ---------------------------------------------------------------- write results
  CSDO_LANES(t) {
    double* so = B.sol + (ad.out_off + t) * 6;
    CSDO_FOR(k, 6, { so[k] = CD(C_SOL + k, t); });
    double* co = B.corr + (ad.out_off + t) * 8;
    CSDO_FOR(k, 4, {
      co[2 * k] = CD(C_CLB + k, t);
      co[2 * k + 1] = CD(C_CUB + k, t);
    });
  }
#if defined(CSDO_PROFILE_PHASES)
  CSDO_PHASE(0);
  if (threadIdx.x == 0 && B.prof) {
    for (int k = 0; k < 16; ++k) B.prof[(int64_t)agent * 48 + k] = prof_acc[k];
    for (int k = 16; k < 24; ++k) B.prof[(int64_t)agent * 48 + 8 + k] = prof_acc[k];
    for (int k = 0; k < 6; ++k) B.prof[(int64_t)agent * 48 + 40 + k] = sub_acc[k];
  }
  if constexpr (ROLE == ROLE_SOLVER) {
    const int ts = CSDO_TID - (int)(blockDim.x >> 1);
    if (B.prof && ts > 0 && ts < Nt && (ts & (ts - 1)) == 0) {   // lanes 1, 2, 4, ...: eliminated at level log2(ts)
      int lv = 0;
      while ((1 << lv) < ts) ++lv;
      B.prof[(int64_t)agent * 48 + 16 + lv] = lvl_fwd;
      B.prof[(int64_t)agent * 48 + 32 + lv] = lvl_bwd;
    }
    if constexpr (MODE != 3) {
      if (B.prof && (ts == 1 || ts == 129)) for (int k = 0; k < 8; ++k) B.prof[(int64_t)agent * 48 + (ts == 1 ? 16 : 32) + k] = xs_acc[k];
    }
  }
#endif
  out.sqp_iters = it;
  out.admm_iters = admm_total;
  out.last_status = status;
#undef ROW
#undef PC_L
#undef PC_G
#undef PROW
}

#undef RZ
#undef RZ_TIME
#undef SH
#undef SU
#undef SX
#undef FE
#undef ER
#undef SINV
#undef CD
#undef WS

}  // namespace csdo
