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
#include "dsqp_layout.h"

// (The experiment switches of rounds 3 - 5 - CSDO_ABSORB_BY_NEIGHBOUR, CSDO_TAIL_GROUPS, CSDO_RUIZ_PARK, CSDO_ONCE_LOOPS, CSDO_TS_LDS, CSDO_TID_*,
//  CSDO_TRIG_CALL, CSDO_BOX_CALL, CSDO_GROW_PACKED, CSDO_SINV_LDS, CSDO_PRIO_ROW - are gone: the winning branch is the code, what lost is
//  recorded with its numbers in DESIGN section 3 and as patches / notes under scripts/experiments/.)
// One-trip loops around the two level-1 steps of the pair-split solve: outside any loop of the iteration the level-1 block is what the
// register allocator spills (18 reloads per solve without them); as do-while - no zero-trip path, so what the step defines needs no merge
// with "what was there before" (as `for` loops: 42 register copies per iteration).  The lane-serial build has no allocation to steer.
#if !defined(CSDO_LANE_MODE_DEVICE)
#define CSDO_ONCE_LOOP
#define CSDO_ONCE_END
#else
#define CSDO_ONCE_LOOP { int once_ = 0; do
#define CSDO_ONCE_END while (++once_ < csdo_opaque_s(1)); }
#endif

#if defined(CSDO_LANE_MODE_DEVICE)
#include <hip/hip_runtime.h>
#define CSDO_FN __device__ __forceinline__
#define CSDO_NOINLINE __device__ __noinline__
// Two specialised lanes per timestep: row lanes are threads [0, Nt), solver lanes are threads [HALF, HALF + Nt).
// The program is instantiated once per role (ROLE_ROW for the first half of the workgroup, ROLE_SOLVER for the
// second): blocks of the other role are compiled out, so each instantiation only carries its own register state.
// Both instantiations execute the same barrier sequence and derive every uniform control value (iteration counts,
// status, rho, norms) from the same LDS broadcasts.
// The lane index of every lanes-block passes through an empty asm: the compiler can then neither hoist the per-field
// workspace / LDS addresses of a block to the top of the program nor keep them alive across blocks (it did: hundreds of
// 64-bit addresses in vector registers, spilled, and reloaded from scratch in front of the accesses).
__device__ __forceinline__ int csdo_opaque(int v) {
  asm volatile("" : "+v"(v));
  return v;
}
__device__ __forceinline__ int csdo_opaque_s(int v) {   // the same for a block-uniform value (stays in a scalar register)
  asm volatile("" : "+s"(v));
  return v;
}
#define csdo_keep(v) csdo_opaque(v)   // a loaded value the compiler must not re-load (rematerialise) at its uses
__device__ __forceinline__ double csdo_keep_f64(double v) {   // the same for a double: also keeps two loads from being merged
  asm volatile("" : "+v"(v));
  return v;
}
// c ? 1.0 : d, selected on the two halves with literal operands.  As a select of doubles the constant 1.0 is hoisted into a
// register pair for the whole kernel, spilled, and reloaded from scratch - with a full wait - in front of every use: the
// sixteen row factors of a Ruiz pass were sixteen scratch round trips in a row.
__device__ __forceinline__ double csdo_one_if(const bool c, const double d) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(d);
  unsigned one_hi;   // (a literal in the source is put back together with the other half into that very select, or shares the register)
  asm volatile("s_mov_b32 %0, 0x3ff00000" : "=s"(one_hi));
  const unsigned lo = c ? 0u : (unsigned)u, hi = c ? one_hi : (unsigned)(u >> 32);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// The thread's index in the workgroup WITHOUT its work-item id register: wave base (a block-uniform value per wave, Shm::wave0: a
// scalar register) + lane number (v_mbcnt: two vector instructions).  As `threadIdx.x` the index lived in v0 for the whole program,
// i.e. it was spilled, and every lanes-block of the cold phases began with a scratch round trip for it - load, wait, compare with
// Nt - before its first useful access (six per Ruiz pass, hundreds per SQP iteration).  CSDO_TID recomputes it where it is asked for
// (volatile: neither hoisted nor kept across blocks - the same job the empty asm of csdo_opaque did for the old form);
// CSDO_TID_HOT (the blocks of the ADMM iteration) is the same.
__device__ __forceinline__ int csdo_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}
#define CSDO_TID (sh.wave0 + csdo_lane_id())
#define CSDO_TID_HOT (sh.wave0 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)))
#define CSDO_LANES(t) if constexpr (ROLE != ROLE_SOLVER) if (const int t = CSDO_TID; t < Nt)
// ROLE_BOTH on the device: one thread per timestep plays both roles (256 threads, 512 registers per lane)
#define CSDO_SOLVER_BASE ((ROLE == ROLE_BOTH) ? 0 : (int)(blockDim.x >> 1))
#define CSDO_SLANES(t) \
  if constexpr (ROLE != ROLE_ROW) if (const int t = CSDO_TID - CSDO_SOLVER_BASE; t >= 0 && t < Nt)
// tail lanes: solver threads [base, base + n_tail), independent of Nt (n_tail <= 36)
#define CSDO_TLANES(t) \
  if constexpr (ROLE != ROLE_ROW) if (const int t = CSDO_TID - CSDO_SOLVER_BASE; t >= 0 && t < n_tail)
// every thread of the solver half, whatever Nt (element-parallel work: t in [0, nthr))
#define CSDO_STHREADS(t, nthr)                                                              \
  if constexpr (ROLE != ROLE_ROW)                                                           \
    if (const int nthr = (ROLE == ROLE_BOTH) ? (int)blockDim.x : (int)(blockDim.x >> 1); true) \
      if (const int t = CSDO_TID - CSDO_SOLVER_BASE; t >= 0)
// blocks of the ADMM iteration keep the plain lane index: their addresses are few and live in registers for the whole
// block of iterations
#define CSDO_LANES_HOT(t) if constexpr (ROLE != ROLE_SOLVER) if (const int t = CSDO_TID_HOT; t < Nt)
#define CSDO_SLANES_HOT(t) \
  if constexpr (ROLE != ROLE_ROW) if (const int t = CSDO_TID_HOT - CSDO_SOLVER_BASE; t >= 0 && t < Nt)
#define CSDO_TLANES_HOT(t) \
  if constexpr (ROLE != ROLE_ROW) if (const int t = CSDO_TID_HOT - CSDO_SOLVER_BASE; t >= 0 && t < n_tail)
// the same 36 lanes taken from the LAST wave of the workgroup: with Nt <= BLOCK/2 - 64 that wave owns no timestep, so what it
// does runs beside the other solver waves' work instead of in front of it
#define CSDO_TLANES_TOP(t) \
  if constexpr (ROLE != ROLE_ROW) if (const int t = CSDO_TID_HOT - ((int)blockDim.x - 64); t >= 0 && t < n_tail)
#define CSDO_STHREADS_HOT(t, nthr)                                                          \
  if constexpr (ROLE != ROLE_ROW)                                                           \
    if (const int nthr = (ROLE == ROLE_BOTH) ? (int)blockDim.x : (int)(blockDim.x >> 1); true) \
      if (const int t = CSDO_TID_HOT - CSDO_SOLVER_BASE; t >= 0)
// Pair-split solve (agent_program, residency modes 0 and 1): EVERY thread of the solver half takes part, whatever Nt - the
// cross-lane moves below need whole waves.  A block of such lanes is a sequence of steps; what one step hands to other lanes
// (through the DPP moves of CSDO_XGET or through LDS) is read in the next one.  On the device the steps are straight-line code of
// one wave (lock step; LDS instructions of a wave complete in order, so an in-wave exchange through LDS needs no barrier); the
// lane-serial build closes the loop over the lanes at every CSDO_XSTEP and opens the next one.
#define CSDO_XLANES(t) if constexpr (ROLE != ROLE_ROW) if (const int t = CSDO_TID - CSDO_SOLVER_BASE; t >= 0)
// lanes 0..5 of every solver wave behind which another wave starts at node s (< Nt): one component q each (sum of the partials that node takes from this wave)
#define CSDO_HANDOVER_LANES(s, q) \
  if constexpr (ROLE != ROLE_ROW) if (const int tt_ = CSDO_TID - CSDO_SOLVER_BASE, q = tt_ & 63, s = tt_ - q + 64; tt_ >= 0 && q < 6 && s < Nt)
#define CSDO_XSTEP(t)               /* register hand-over (DPP): program order is all it needs */
#define CSDO_XSTEP_LDS(t) csdo_wave_sync();   /* in-wave hand-over through LDS */
__device__ __forceinline__ void csdo_wave_sync() {   // orders the wave's own LDS stores before its later LDS loads for the compiler
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// DPP controls (v_mov_b32_dpp): the source lane of lane i within its wave of 64
#define CSDO_DPP_PAIR_EVEN 0xA0   /* quad_perm [0,0,2,2]: i & ~1 */
#define CSDO_DPP_PAIR_ODD 0xF5    /* quad_perm [1,1,3,3]: i | 1  */
#define CSDO_DPP_PAIR_SWAP 0xB1   /* quad_perm [1,0,3,2]: i ^ 1  */
#define CSDO_DPP_PREV 0x138       /* wave_shr:1: i - 1, lane 0 reads 0.0 */
template <int CTRL>
__device__ __forceinline__ double csdo_dpp_f64(const double v) {   // (bound_ctrl: a lane without a source reads 0 - and no copy of an old value is made)
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#define CSDO_XGET(CTRL, fld, k) csdo_dpp_f64<CTRL>(V.fld[k])
#define CSDO_LS(t) lanes_r
#define CSDO_SS(t) lanes_s
#define CSDO_SYNC() __syncthreads()
#define CSDO_MARK(name) asm volatile("; CSDO_MARK " name)
// keep the instruction scheduler from merging hand-pipelined stages (it would raise register pressure again)
#define CSDO_STAGE() __builtin_amdgcn_sched_barrier(0)
#if defined(CSDO_PROFILE_PHASES)
// diagnostic build: thread 0 accumulates shader-clock ticks per phase (never enabled in the shipped library)
#define CSDO_PHASE(k)                                                    \
  do {                                                                   \
    if (threadIdx.x == 0) {                                              \
      const long long now_ = (long long)__builtin_amdgcn_s_memtime();    \
      prof_acc[prof_cur] += now_ - prof_last;                            \
      prof_last = now_;                                                  \
      prof_cur = (k);                                                    \
    }                                                                    \
  } while (0)
#else
#if defined(CSDO_ASM_MARKS)
#define CSDO_PHASE(k) asm volatile("; CSDO_MARK phase_" #k)
#else
#define CSDO_PHASE(k) ((void)0)
#endif
#endif
#elif defined(CSDO_LANE_MODE_SERIAL)
#define csdo_keep(v) (v)
#define csdo_keep_f64(v) (v)
#define csdo_opaque_s(v) (v)
#define csdo_one_if(c, d) ((c) ? 1.0 : (d))
#define CSDO_FN inline
#define CSDO_NOINLINE inline
#define CSDO_LANES(t) if constexpr (ROLE != ROLE_SOLVER) for (int t = 0; t < Nt; ++t)
#define CSDO_SLANES(t) if constexpr (ROLE != ROLE_ROW) for (int t = 0; t < Nt; ++t)
#define CSDO_TLANES(t) if constexpr (ROLE != ROLE_ROW) for (int t = 0; t < n_tail; ++t)
#define CSDO_STHREADS(t, nthr) \
  if constexpr (ROLE != ROLE_ROW) if (const int nthr = 64; true) for (int t = 0; t < nthr; ++t)
#define CSDO_LANES_HOT(t) CSDO_LANES(t)
#define CSDO_SLANES_HOT(t) CSDO_SLANES(t)
#define CSDO_TLANES_HOT(t) CSDO_TLANES(t)
#define CSDO_TLANES_TOP(t) CSDO_TLANES(t)
#define CSDO_STHREADS_HOT(t, nthr) CSDO_STHREADS(t, nthr)
#define CSDO_XLANES(t) if constexpr (ROLE != ROLE_ROW) for (int t = 0; t < NtE; ++t)
#define CSDO_HANDOVER_LANES(s, q) if constexpr (ROLE != ROLE_ROW) for (int s = 64; s < Nt; s += 64) for (int q = 0; q < 6; ++q)
#define CSDO_XSTEP(t) } CSDO_XLANES(t) { SolvRegs& V = CSDO_SS(t);
#define CSDO_XSTEP_LDS(t) CSDO_XSTEP(t)
#define CSDO_DPP_PAIR_EVEN 0xA0
#define CSDO_DPP_PAIR_ODD 0xF5
#define CSDO_DPP_PAIR_SWAP 0xB1
#define CSDO_DPP_PREV 0x138
// what v_mov_b32_dpp does with these controls: source lane of lane t (waves of 64), -1: none (reads 0.0)
inline int csdo_dpp_src(const int ctrl, const int t, const int n_lanes) {
  const int lane = t & 63, base = t - lane;
  int src = -1;
  if (ctrl < 0x100) src = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
  else if (ctrl == 0x138) src = lane - 1;
  else if (ctrl == 0x130) src = lane + 1 < 64 ? lane + 1 : -1;
  else if (ctrl > 0x100 && ctrl <= 0x10F) src = ((lane & 15) + (ctrl & 15) < 16) ? lane + (ctrl & 15) : -1;
  if (src < 0 || base + src >= n_lanes) return -1;
  return base + src;
}
#define CSDO_XGET(CTRL, fld, k) (csdo_dpp_src(CTRL, t, NtE) >= 0 ? lanes_s[csdo_dpp_src(CTRL, t, NtE)].fld[k] : 0.0)
#define CSDO_LS(t) lanes_r[t]
#define CSDO_SS(t) lanes_s[t]
#define CSDO_SYNC() ((void)0)
#define CSDO_MARK(name) ((void)0)
#define CSDO_STAGE() ((void)0)
#define CSDO_PHASE(k) ((void)0)
#else
#error "define CSDO_LANE_MODE_DEVICE or CSDO_LANE_MODE_SERIAL"
#endif

#include "csdo_math.h"   // sin, cos, tan, atan2: one source for the device build and the lane-serial build (same bits)

namespace csdo {

// The program's call sites of the shared trigonometry.  On the device they are REAL CALLS: every
// inlined copy of a cold piece moves the register allocation of the ADMM loop - inlined at their eight
// sites these functions put six scratch reloads into every solve of the 512-thread class and the map100 step went from 60.2
// to 69.4 ms; as calls (doubles in, doubles out: nothing lives in memory across them) the program has fewer spills than before
// (VGPR dwords 429 -> 309, SGPR 885 -> 647) and the step is 59.8 ms.  Same bits either way.
struct SinCos { double s, c; };
#if defined(CSDO_LANE_MODE_DEVICE)
#define CSDO_TRIG_FN CSDO_NOINLINE
#else
#define CSDO_TRIG_FN CSDO_FN
#endif
CSDO_TRIG_FN SinCos sincos_of(const double x) {
  SinCos r;
  xm::sincos(x, r.s, r.c);
  return r;
}
CSDO_TRIG_FN double tan_of(const double x) { return xm::tan(x); }
CSDO_TRIG_FN double atan2_of(const double y, const double x) { return xm::atan2(y, x); }

constexpr int ROLE_BOTH = 0, ROLE_ROW = 1, ROLE_SOLVER = 2;   // ROLE_BOTH: lane-serial host build

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
// BCR stops when at most TAIL_NODES nodes remain; the remaining block-tridiagonal system (<= 36 unknowns) is solved
// with its explicit dense inverse by 6*R lanes in one phase instead of log2(R)+1 forward and backward level phases.
// (TAIL_NODES, TAIL_N, TAIL_NODES_BIG, TAIL_N_BIG: dsqp_layout.h)

CSDO_FN constexpr int row_col(int i, int s) {
  constexpr int T[NROW][3] = {{0, 2, 4}, {1, 2, 4}, {2, 3, 4}, {3, 5, -1}, {0, -1, -1}, {1, -1, -1},
                              {2, -1, -1}, {0, 2, -1}, {1, 2, -1}, {0, 2, -1}, {1, 2, -1}, {0, -1, -1},
                              {1, -1, -1}, {4, -1, -1}, {5, -1, -1}, {3, -1, -1}};
  return T[i][s];
}
// Ruiz equilibration: the coefficients (row, slot) that wait in LDS between the passes (dsqp_program_impl.h, CSDO_RUIZ_PARK), and where
CSDO_FN constexpr int ruiz_park_slot(int i, int s) {
  if (s == 0 && i >= 11 && i <= 15) return i - 11;       // trust x, y; |v|, |w|; |steer|: their single coefficient
  if (s == 1 && i >= 7 && i <= 10) return 5 + (i - 7);   // corridor rows: the yaw coefficient
  return -1;
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

// Block-uniform values fetched with vector loads (per-agent descriptors) would sit in VGPRs: move them to SGPRs.
#if defined(CSDO_LANE_MODE_DEVICE)
CSDO_FN int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
CSDO_FN long long uniform_i64(long long v) {
  const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffll));
  const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
  return ((long long)hi << 32) | (unsigned int)lo;
}
CSDO_FN double uniform_f64(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <class T>
CSDO_FN T* uniform_ptr(T* p) { return (T*)uniform_i64((long long)p); }
// An LDS pointer that went through a function argument has lost its address space (flat accesses); rebuild it from the
// workgroup's dynamic LDS symbol so that the compiler emits ds_* instructions again.
CSDO_FN double* lds_ptr(double* p) {
  extern __shared__ __align__(16) double csdo_lds_base[];
  const int off = uniform_i32((int)(p - (double*)csdo_lds_base));
  return csdo_lds_base + off;
}
#else
CSDO_FN double* lds_ptr(double* p) { return p; }
template <class T>
CSDO_FN T* uniform_ptr(T* p) { return p; }
CSDO_FN int uniform_i32(int v) { return v; }
CSDO_FN long long uniform_i64(long long v) { return v; }
CSDO_FN double uniform_f64(double v) { return v; }
#endif

CSDO_FN double dmax(double a, double b) { return (b > a) ? b : a; }   // NaN in b is ignored, like vec_norm_inf
CSDO_FN double dmin(double a, double b) { return (b < a) ? b : a; }
CSDO_FN double osqp_max(double a, double b) { return (a > b) ? a : b; }  // c_max
CSDO_FN double osqp_min(double a, double b) { return (a < b) ? a : b; }  // c_min
CSDO_FN int osqp_min_i(int a, int b) { return (a < b) ? a : b; }
// the projections of the ADMM iteration: one v_max_f64 / v_min_f64 each instead of a compare and two selects per c_max / c_min
// (same values for ordered operands; a NaN iterate is replaced by the bound by both forms)
CSDO_FN double hot_max(double a, double b) { return __builtin_fmax(a, b); }
CSDO_FN double hot_min(double a, double b) { return __builtin_fmin(a, b); }
CSDO_FN double limit_scaling(double d) {
  d = csdo_one_if(d < MIN_SCALING, d);
  d = d > MAX_SCALING ? MAX_SCALING : d;
  return d;
}
// maxima of absolute values (accumulator a >= 0, never NaN; a NaN in b is ignored like dmax does): one v_max_f64
CSDO_FN double nmax(double a, double b) { return __builtin_fmax(a, b); }
// limit_scaling of such a maximum (d >= 0, never NaN): the upper clamp as one v_min_f64
CSDO_FN double limit_norm(double d) {
  d = csdo_one_if(d < MIN_SCALING, d);
  return __builtin_fmin(d, MAX_SCALING);
}
// 1.0 / sqrt(x) for x = limit_norm(.) in [1e-4, 1e4]: the compiler's own expansions of the IEEE square root and division (same
// instructions, same order: same bits) without what they carry for arguments that cannot occur here - the rescaling of tiny
// or huge operands (v_ldexp / v_div_scale, identities in this range) and the special-case selects (v_cmp_class, v_div_fixup):
// 18 instead of 35 instructions, and a Ruiz pass computes 22 of them per timestep.
#if defined(CSDO_LANE_MODE_DEVICE)
CSDO_FN double inv_sqrt_limited(const double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  double d = fma(-g, g, x);
  h = fma(h, r, h);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);                       // g = sqrt(x)
  double q = __builtin_amdgcn_rcp(g);
  double e = fma(-g, q, 1.0);
  q = fma(q, e, q);
  e = fma(-g, q, 1.0);
  q = fma(q, e, q);
  const double rr = fma(-g, q, 1.0);      // (numerator 1.0: the quotient estimate is q itself)
  return fma(rr, q, q);
}
#else
CSDO_FN double inv_sqrt_limited(const double x) { return 1.0 / sqrt(x); }
#endif
CSDO_FN constexpr int sym(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }

// ---------------------------------------------------------------------------------------------------------
// Per-lane state (registers on the device)
// ---------------------------------------------------------------------------------------------------------
// Register state of the two lane roles (each program instantiation only ever touches its own).  Both are caches of the per-agent workspace: loaded at the start of a block of <= check_termination ADMM iterations
// and written back at its end; every cold phase works from the workspace with short-lived temporaries.
// How much of a BCR node's factor (F_l 36 + F_r 36 + packed pivot inverse 21 = 93 doubles) its solver lane keeps in
// registers during an ADMM block: F_l and the first ER_REG entries of F_r.  With everything in registers (198 of the 256
// the 512-thread kernel has per lane) the allocator spills a third of it to scratch, and a scratch reload inside a BCR
// level costs more than the LDS read it replaced (measured: 160 ms against 140 ms per step on the map100 set); 60 doubles
// leave the hot loop spill-free.  The other 33 sit in LDS (Shm::fx): read by one lane, once per sweep.
static_assert(ER_REG1 <= ER_REG, "SolvRegs::er is sized for mode 0");
// The 768-thread class (residency mode 2, 168 registers per lane): NOTHING of the factor stays in registers across an iteration -
// whatever part was declared lane state there, the allocator spilled and reloaded level by level.  The lane's level-1 block is
// fetched from the workspace in one batch in front of each of its two uses, the block of its other level sits in LDS, all 36
// entries (Shm::fx, LD_fx2 = 34 doubles per lane - 2 x odd: conflict-free 128-bit reads - and the two doubles per lane that
// Shm::carry does not use); to make room the rows' rhs shares use the right partials' array (see Shm::rhs).
// 18 + 34 = 52 doubles per timestep: horizons to 350 beside 238 obstacles; longer ones run mode 3.
struct RowRegs {            // row lane of timestep t: the 16 home constraint rows and the 6 variables
  double c[NROW][3];        // scaled coefficients on own columns
  double cn[4];             // scaled coefficient of kin rows on column i of t+1
  double lo[NROW], hi[NROW];// scaled bounds
  double y[NROW], z[NROW];  // ADMM dual / slack (z doubles as Ruiz scratch before the warm start)
  double x[6];              // scaled primal iterate
  double b[6];              // Ruiz scratch / rhs temporary
  double Pvv, Pww, Pvn;     // scaled objective (set-up stage; kept in the workspace afterwards)
  unsigned eqmask;          // rows in OSQP's "equality" class (rho * 1e3)
  unsigned loosemask;       // rows with both bounds infinite (rho = RHO_MIN)
  unsigned act;             // rows that exist at this t
  int ncols;                // 6, or 4 at t = Nt-1
};
struct SolvRegs {           // solver lane of timestep t during an ADMM block
  double b[6];              // rhs -> BCR work vector -> x_tilde
  // Mode 3 (horizons beyond 384, or a 768-thread agent whose obstacles leave no room): BCR node t with both couplings, F_l = Sinv E_l
  // in el, F_r fetched from the workspace where it is used.
  // Pair-split modes (0, 1, 2): the two 6x6 blocks this LANE multiplies with, one per level it works at (see "pair-split solve"
  // in dsqp_program_impl.h): el = the block of level 1, er = the first ER_REG (mode 2: ER_REG2) entries of the block of the lane's
  // level >= 2 (the rest: LDS, Shm::fx)
  double el[36];
  int ts0, ts1;             // plane range of this timestep (CSR offsets), for the rhs assembly
  double er[ER_REG];
  double v[6], o[6];        // pair-split: operand and product of the level at hand (short-lived; lane state only for the lane-serial build)
  double b0[6], x0[6], rl[6];   // REFINE kernels only: the iteration's rhs, the first solve's x~ (1), the lagged residual (2); dead code elsewhere
  unsigned fl;              // pair-split: what the lane does at which level (XF_* below)
};
typedef RowRegs LaneState;  // the row lane is the "home" of a timestep

// Per-agent workspace in HBM (L2-resident), SoA [slot][stride]: the master copy of every per-lane quantity.
enum WsSlot {
  W_C = 0,      // [48] c[i][s] at 3*i+s
  W_CN = 48,    // [4]
  W_LO = 52,    // [16]
  W_HI = 68,    // [16]
  W_Yv = 84,    // [16]
  W_Zv = 100,   // [16]
  W_X = 116,    // [6]
  W_SINV = 122, // [21]
  W_P = 143,    // [3] Pvv, Pww, Pvn
  C_E = 146,    // [16] Ruiz row scaling
  C_D = 162,    // [6]  Ruiz column scaling
  C_SOL0 = 168, // [6]  previous SQP iterate == linearisation point (x,y,yaw,steer,v,w), unscaled
  C_SOL = 174,  // [6]  latest QP solution, unscaled
  C_XT = 180, C_YT = 181, C_YAWT = 182,  // original initial guess (trust centre, start/goal rows): never moved
  C_CLB = 183,  // [4]  corridor of the linearisation point: xf, yf, xr, yr lower
  C_CUB = 187,  // [4]  upper
  C_DY = 191,   // [16] last y-increment, kept on termination-check iterations only
  W_ACT = 207, W_EQ = 208, W_LOOSE = 209,  // row masks as exactly-representable doubles (read by the solver lanes)
  C_TOTAL = 210
};

// lane-major leading dimensions (doubles per lane) of the LDS arrays.  All are 2 * odd: 16-byte aligned lanes, and the
// ds_read_b128 / ds_write_b128 of 16 consecutive lanes (also of lanes a power of two apart) fall into 16 different
// 4-bank groups - conflict free.  (12 doubles, the former reduction stride, is 2-way conflicting.)
static_assert(LD_block2 >= LD_stash, "the factorisation parks a 6x6 product and the packed pivot inverse per lane in the block's arrays");

// Shared (LDS) arrays, lane-major: element k of lane t at arr[t * LD + k]
struct Shm {
  double* vec;      // [stride][6]  x_tilde / x exchange
  double* pl;       // [stride][6]  BCR partials for the left neighbour (forward sweep only); ALIASES vec: a node's rhs is
                    //              in registers before it publishes its partial, and its x is written in the backward sweep
  double* pr;       // [stride][6]  BCR partials for the right neighbour  (solve only)
  double* rhs;      // [stride][6]  the row lane's own share of the next rhs: sigma x + A'(rho z - y) of its home rows
  double* carry;    // [stride][6]  t -> t+1 hand-over (kinematic rows' share of the next rhs; norms in the set-up stage)
  double* carry2;   // [stride][6]  t -> t-1 hand-over (set-up stage, update_info, feasibility test); ALIASES lohi in modes 0, 1
                    //              (only used between ADMM blocks; the bounds are reloaded at a block's start)
  double* lohi;     // [stride][22] bounds of the home rows during an ADMM block: 0..6 eq rows (lo = hi),
                    //              7..12 lo and 13..18 hi of corridor/trust rows, 19..21 hi of the +-boxes (lo = -hi)
  double* red;      // [stride][14] reduction scratch; ALIASES vec/pr/rhs (reductions only run between ADMM blocks)
  double* fx;       // [stride][34] the part of a BCR node's factor that does not fit its solver lane's registers during an
                    //              ADMM block: F_r[24..36) and the pivot-block inverse (packed lower, 21)
  double* stash;    // [stride][38] factor-time scratch of an eliminated node (one 6x6 product); ALIASES everything from vec
                    //              on: the factorisation runs between blocks, bounds and rhs are (re)loaded at a block's start
  double* obs;      // [3][n_obs]
  double* facE;     // [stride][72] F_l (36) + F_r (36) of every node, lane-major (global; in solver registers during a block)
  double* facX;     // [100][stride] factor-time exchange and the nodes' diagonal blocks (global, coalesced)
  double* cold;     // [C_TOTAL][stride] per-agent workspace (global)
  double* bcast;    // [32] block-wide results
  double* tinv;     // [36][38] dense inverse of the BCR tail system, one row per tail lane (BIGT: [6 tn][6 tn + 2])
  double* tvec;     // [2][36] tail rhs gather / Gauss-Jordan pivot row (BIGT: [2][72])
  int ld_tinv;      // BIGT kernels: row stride of tinv, 6 * tail_nodes + 2
  int tvec_half;    // BIGT kernels: offset of tvec's second half (36, or 72 for a tail of more than six nodes)
  double* pc;       // [K][3]  per-plane share of A'(rho z - y) for the next rhs, LDS copy (agents whose planes fit: rows_lds)
  double* pcg;      // [K][3]  the same in the workspace (all other agents)
  double* prow;     // [K][10] rows_lds: duals, slacks (y[4], z[4]) and timestep of a plane's four inter-vehicle rows during
                    //              an ADMM block
  double* pco;      // [16][n_pco_ld] rows_lds: coefficients a, b, c_yaw and upper bounds of the four rows of the first n_pco planes
                    // during an ADMM block, in whatever LDS the launch has left (read-only there; from the workspace they
                    // were an HBM round trip in every iteration's plane pass)
  int n_pco, n_pco_ld;
  int stride;
  int wave0;        // device: index of the wave's first thread in the workgroup (block-uniform per wave; see CSDO_TID)
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

// inter-vehicle row workspace: SoA by field, [field][4K] per agent (the 4 rows of a plane are contiguous)
enum { R_Y = 0, R_Z = 1, R_DY = 2, R_U = 3, R_E = 4, R_CA = 5, R_CB = 6, R_CY = 7 };   // then [3K] plane shares (MODE 3)

// =========================================================================================================
// Block-wide reductions.  Partials are stored per lane in sh.pl/sh.pr (12 slots); after a barrier the first wave
// folds them.  max is order independent; sums use a fixed order (stride-64 serial, then a halving tree) that the
// serial build reproduces exactly, so device and emulation agree bit for bit.
// =========================================================================================================
template <int K>
CSDO_FN void red_put(const Shm& sh, int t, const double (&part)[K]) {
  static_assert(K <= 12, "reduction scratch holds 12 values per lane");
  CSDO_FOR(k, K, { sh.red[t * LD_red + k] = part[k]; });
}

#if defined(CSDO_LANE_MODE_DEVICE)
CSDO_FN double wave_shfl_down(double v, int off) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_down(lo, off, 64);
  hi = __shfl_down(hi, off, 64);
  return __hiloint2double(hi, lo);
}
#endif

#if defined(CSDO_LANE_MODE_DEVICE)
// the value of lane + 1 of the wave (DPP wave_shl:1, supported on gfx950; the last lane reads 0)
CSDO_FN double wave_next(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x130, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x130, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
#endif

#if defined(CSDO_LANE_MODE_DEVICE)
CSDO_FN double wave_shfl_xor(double v, int mask) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __shfl_xor(lo, mask, 64);
  hi = __shfl_xor(hi, mask, 64);
  return __hiloint2double(hi, lo);
}
#endif

// out[k] = reduction over lanes 0..Nt-1 of slot k.  IS_SUM=false: max with NaNs ignored.  Collective.
// Maxima (order independent) take the wide path: the first wave's lane (k, segment) scans every fourth lane's slot k and
// two butterfly steps join the four segments - the slot-after-slot fold with a six-step shuffle tree per slot was 15 k cycles
// for the twelve norms of update_info.  Sums keep the fixed order that the serial build reproduces.
template <int K, bool IS_SUM>
CSDO_FN void red_fold(const Shm& sh, int Nt, double (&out)[K]) {
  CSDO_SYNC();
#if defined(CSDO_LANE_MODE_DEVICE)
  const int tid = CSDO_TID;
  if constexpr (!IS_SUM && K > 2) {
    static_assert(K <= 16, "one 16-lane row per segment");
    if (tid < 64) {
      const int k = tid & 15, seg = tid >> 4;
      double acc = 0.0;
      if (k < K) {
#pragma unroll 8
        for (int j = seg; j < Nt; j += 4) acc = dmax(acc, sh.red[j * LD_red + k]);
      }
      acc = dmax(acc, wave_shfl_xor(acc, 16));
      acc = dmax(acc, wave_shfl_xor(acc, 32));
      if (tid < K) sh.bcast[tid] = acc;
    }
    __syncthreads();
    CSDO_FOR(k, K, { out[k] = uniform_f64(sh.bcast[k]); });
    __syncthreads();
    return;
  }
  if (tid < 64) {
    CSDO_FOR(k, K, {
      double acc = 0.0;
      for (int j = tid; j < Nt; j += 64) {
        const double v = sh.red[j * LD_red + k];
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
  // every lane reads the same LDS word: hand the result over in scalar registers (a block-uniform double that stays in
  // vector registers - the residual norms, rho - costs two of them for as long as it lives, across whole ADMM blocks)
  CSDO_FOR(k, K, { out[k] = uniform_f64(sh.bcast[k]); });
  __syncthreads();
#else
  CSDO_FOR(k, K, {
    double lane[64];
    for (int l = 0; l < 64; ++l) {
      double acc = 0.0;
      for (int j = l; j < Nt; j += 64) {
        const double v = sh.red[j * LD_red + k];
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

// *addr = max(*addr, v) for non-negative finite doubles held in LDS, from any thread of the workgroup: the bit patterns of
// non-negative doubles order like unsigned integers, so this is one ds_max_u64.  NaN and zero leave the maximum alone (dmax).
CSDO_FN void lds_max_nonneg(double* addr, double v) {
  if (v > 0.0) {
#if defined(CSDO_LANE_MODE_DEVICE)
    atomicMax((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
#else
    if (v > *addr) *addr = v;
#endif
  }
}

// Sum over lanes 0..Nt-1 of the field-major array field[k * stride + t], in red_fold's order (stride-64 serial, then a
// halving tree).  Collective; the caller has a barrier between the writes of the field and this call.
CSDO_FN void field_sum(const Shm& sh, int Nt, int k, double (&out)[1]) {
  const double* f = sh.vec + (size_t)k * (size_t)sh.stride;
#if defined(CSDO_LANE_MODE_DEVICE)
  const int tid = CSDO_TID;
  if (tid < 64) {
    double acc = 0.0;
    for (int j = tid; j < Nt; j += 64) acc = acc + f[j];
    for (int off = 32; off >= 1; off >>= 1) acc = acc + wave_shfl_down(acc, off);
    if (tid == 0) sh.bcast[0] = acc;
  }
  __syncthreads();
  out[0] = uniform_f64(sh.bcast[0]);
  __syncthreads();
#else
  double lane[64];
  for (int l = 0; l < 64; ++l) {
    double acc = 0.0;
    for (int j = l; j < Nt; j += 64) acc = acc + f[j];
    lane[l] = acc;
  }
  for (int off = 32; off >= 1; off >>= 1)
    for (int l = 0; l < off; ++l) lane[l] = lane[l] + lane[l + off];
  out[0] = lane[0];
#endif
}

// =========================================================================================================
// Safe-corridor boxes (sqp/corridor.cc of the reference), one lane per point.
// =========================================================================================================
struct BoxD {
  double x_min, y_min, x_max, y_max;
};

// Obstacles that can ever matter for boxes grown from (xc, yc): a box never extends more than 10.1 m from its seed
// (101 steps of 0.1 m, corridor.cc:305), so an obstacle whose inflated square cannot reach that extent fails at least
// one of the four strict inequalities for every box of this seed.  The 0.5 m slack dwarfs any rounding, so skipping
// those obstacles is exact.  Up to 256 obstacles as four 64-bit masks held in registers.
struct ObsMask {
  unsigned long long m[4];
};
constexpr int OBS_MASK_CAP = 256;

// One walk over the obstacles for both questions about a point: which obstacles can ever touch a box grown from it (the
// mask), and - make_box's isPointCollision - the first obstacle whose inflated square contains it (`hit`, -1: none).  Four
// obstacles per trip: their twelve LDS reads are in flight together (one obstacle per trip was an LDS round trip per obstacle and
// walk, twice per box: the longest part of the corridor phase).
CSDO_FN ObsMask cull_obstacles(double xc, double yc, const double* obs, int n_obs, double rv, int& hit) {
  ObsMask M;
  M.m[0] = M.m[1] = M.m[2] = M.m[3] = 0ull;
  const double reach = 10.1 + 0.5;
  int first = -1;
  auto one = [&](const int k, const double ox, const double oy, const double ri) __attribute__((always_inline)) {
    // (ri = r_obs + rv, as staged.)  The point test only for obstacles that pass the - much wider - cull: five vector
    // instructions per obstacle otherwise
    const double infl = ri + reach;
    const double dx = ox - xc, dy = oy - yc;
    if (fabs(dx) < infl && fabs(dy) < infl) {   // == dx > -infl && dx < infl && dy > -infl && dy < infl
      if (first < 0 && (xc - ri) < ox && ox < (xc + ri) && (yc - ri) < oy && oy < (yc + ri)) first = k;
      if (k < OBS_MASK_CAP) {
        const unsigned long long bit = 1ull << (k & 63);
        if (k < 64) M.m[0] |= bit;
        else if (k < 128) M.m[1] |= bit;
        else if (k < 192) M.m[2] |= bit;
        else M.m[3] |= bit;
      }
    }
  };
  int k = 0;
  for (; k + 4 <= n_obs; k += 4) {
    double ox[4], oy[4], r[4];
    CSDO_FOR(q, 4, {
      ox[q] = obs[k + q];
      oy[q] = obs[n_obs + k + q];
      r[q] = obs[2 * n_obs + k + q];
    });
    CSDO_FOR(q, 4, { one(k + q, ox[q], oy[q], r[q]); });
  }
  for (; k < n_obs; ++k) one(k, obs[k], obs[n_obs + k], obs[2 * n_obs + k]);
  hit = first;
  return M;
}

CSDO_FN bool obstacle_in_box(const BoxD& b, const double* obs, int n_obs, int k, double rv) {
  const double ox = obs[k], oy = obs[n_obs + k], infl = obs[2 * n_obs + k];   // (staged as r_obs + rv)
  return (b.x_min - infl) < ox && ox < (b.x_max + infl) && (b.y_min - infl) < oy && oy < (b.y_max + infl);
}

CSDO_FN int ctz64(unsigned long long v) {
#if defined(CSDO_LANE_MODE_DEVICE)
  return __ffsll((long long)v) - 1;
#else
  return __builtin_ctzll(v);
#endif
}

CSDO_FN bool box_valid(const BoxD& b, const double* obs, int n_obs, double dimx, double dimy, double rv,
                       const ObsMask& M) {
  // isBoxValid, corridor.cc:252-272: inside [rv, dim-rv]^2 and no obstacle centre strictly inside the box
  // inflated by (r_obs + rv) on every side
  if (b.x_min < rv || b.x_max > dimx - rv || b.y_min < rv || b.y_max > dimy - rv) return false;
  bool hit = false;   // (a `return` inside CSDO_FOR would only leave its lambda)
  CSDO_FOR(w, 4, {
    unsigned long long m = M.m[w];
    while (m && !hit) {
      const int k = 64 * w + ctz64(m);
      m &= m - 1;
      hit = obstacle_in_box(b, obs, n_obs, k, rv);
    }
  });
  for (int k = OBS_MASK_CAP; k < n_obs && !hit; ++k) hit = obstacle_in_box(b, obs, n_obs, k, rv);
  return !hit;
}

// generateLocalBox, corridor.cc:278-324: grow by 0.1 in the order +y, -x, -y, +x until blocked or >= 10 m.
// isBoxValid's obstacle test is the conjunction of four strict inequalities, one per box side.
//
// The reference walks 404 trials; done trial by trial on the device that is ~60 VALU instructions per trial whatever the
// surroundings, it saturates the SIMDs (both roles grow boxes at the same time) and was 15-20 % of a short agent's time.
// But the process has very little freedom.  While a side is moving it takes exactly one step per round, so for every
// (obstacle, side) there is a first step count E at which that side's inequality turns true - and it stays true - and a
// trial of side d in round r fails exactly when some obstacle has all four inequalities true with the sides before d in the
// round at r steps and the sides after d at r-1.  For one obstacle that first happens in round max_e E_e, at the LAST side
// (in round order) that attains the maximum; a side that stops earlier with its inequality still false removes the obstacle
// for good.  So the whole growth is four "a side stops" events (blocked by an obstacle, by the map border, or after the
// 101 steps that make len >= 10 in floating point), each found by a minimum over the culled obstacles, followed by a replay
// of the accepted steps so that every coordinate is the same sequence of floating-point additions as in the reference.
// E comes from a quotient; the sequential sums differ from c0 + n * 0.1 by < 1e-12, so the quotient is only trusted when it is
// further than 1e-7 steps from an integer; otherwise E is found by replaying the additions against the exact inequality.
// Results are bit-identical to the trial-by-trial walk (tests/test_boxes_serial.py against the oracle's, random maps).
constexpr int grow_limit_steps() {   // accepted steps until len >= l_limit, len summed as the reference does: 101
  double len = 0.0;
  int n = 0;
  while (!(len >= 10.0)) {
    len += 0.1;
    ++n;
  }
  return n;
}
constexpr int GROW_LIMIT = grow_limit_steps();
constexpr int GROW_NEVER = 120;      // "not within any reachable step count"

// Step count at which a side's inequality against a threshold turns true.  Side d (0: +y, 1: -x, 2: -y, 3: +x) moves the
// coordinate c_n (c_0 = c0, c_n = c_{n-1} +- 0.1, summed sequentially); the inequality is  o < c_n + infl  for the sides that
// move up and  c_n - infl < o  for the sides that move down (isBoxValid's forms; the map border is the same with infl = 0).
// Returns the smallest n >= 0 for which it holds (it is monotone in n), GROW_NEVER if none is in reach.
CSDO_FN int first_step(const int d, const double xc, const double yc, const double ox, const double oy, const double infl) {
  const bool vertical = (d & 1) == 0, up = (d == 0 || d == 3);
  const double c0 = vertical ? yc : xc, o = vertical ? oy : ox;
  const double step = up ? 0.1 : -0.1;
  const double g = (up ? ((o - infl) - c0) : (c0 - (o + infl))) * 10.0;
  if (!(g > -1.0)) return 0;                      // beyond by more than a step already
  if (!(g < (double)GROW_NEVER)) return GROW_NEVER;
  const double fl = floor(g);
  const double frac = g - fl;
  if (frac < 1e-7 || frac > 1.0 - 1e-7) {         // too close to call from the quotient: walk the additions
    double c = c0;
    int n = 0;
    while (n < GROW_NEVER && !(up ? (o < (c + infl)) : ((c - infl) < o))) {
      c += step;
      ++n;
    }
    return n;
  }
  return (g < 0.0) ? 0 : (int)fl + 1;
}

// The four step counts of an obstacle (one per side, each <= GROW_NEVER < 256) do not change from pass to pass: the first pass
// packs them into one word per culled obstacle, the other three read them back instead of running first_step twelve more times
// per obstacle (the corridor phase is the one phase that is bound by instruction issue: both roles grow boxes at once).  The words
// live in whatever memory the caller has idle - the agent program: the ADMM block's LDS arrays, entry j of thread i at
// base[j * stride + i] (conflict free) -; obstacles beyond `cap` are recomputed.  cap = 0: no cache.
struct BoxCache {
  unsigned* base;
  int stride, cap;
};
// first_step with the side known at compile time (no selects on d; the four of an obstacle are straight-line code)
template <int D>
CSDO_FN int first_step_of(const double xc, const double yc, const double ox, const double oy, const double infl) {
  constexpr bool vertical = (D & 1) == 0, up = (D == 0 || D == 3);
  const double c0 = vertical ? yc : xc, o = vertical ? oy : ox;
  const double g = (up ? ((o - infl) - c0) : (c0 - (o + infl))) * 10.0;
  if (!(g > -1.0)) return 0;
  if (!(g < (double)GROW_NEVER)) return GROW_NEVER;
  const double fl = floor(g);
  const double frac = g - fl;
  if (frac < 1e-7 || frac > 1.0 - 1e-7) {
    const double step = up ? 0.1 : -0.1;
    double c = c0;
    int n = 0;
    while (n < GROW_NEVER && !(up ? (o < (c + infl)) : ((c - infl) < o))) {
      c += step;
      ++n;
    }
    return n;
  }
  return (g < 0.0) ? 0 : (int)fl + 1;
}
CSDO_FN unsigned obstacle_steps(const double xc, const double yc, const double ox, const double oy, const double infl) {
  return (unsigned)first_step_of<0>(xc, yc, ox, oy, infl) | ((unsigned)first_step_of<1>(xc, yc, ox, oy, infl) << 8) |
         ((unsigned)first_step_of<2>(xc, yc, ox, oy, infl) << 16) | ((unsigned)first_step_of<3>(xc, yc, ox, oy, infl) << 24);
}
// The culled obstacles in mask order (the order of their cache slots)
struct ObsWalk {
  unsigned long long m0, m1, m2, m3;
  int k_tail;
};
CSDO_FN bool next_obstacle(ObsWalk& w, const int n_obs, int& k) {
  if (w.m0) {
    k = ctz64(w.m0);
    w.m0 &= w.m0 - 1;
  } else if (w.m1) {
    k = 64 + ctz64(w.m1);
    w.m1 &= w.m1 - 1;
  } else if (w.m2) {
    k = 128 + ctz64(w.m2);
    w.m2 &= w.m2 - 1;
  } else if (w.m3) {
    k = 192 + ctz64(w.m3);
    w.m3 &= w.m3 - 1;
  } else if (w.k_tail < n_obs) {
    k = w.k_tail++;
  } else {
    return false;
  }
  return true;
}
// n sequential additions of `step` (the reference's own sums), eight to a trip: a trip per addition was a branch per addition,
// and the wave runs as many trips as its longest lane
CSDO_FN double add_steps(double c, const double step, int n) {
  for (; n >= 8; n -= 8) {
    c += step; c += step; c += step; c += step;
    c += step; c += step; c += step; c += step;
  }
  for (; n > 0; --n) c += step;
  return c;
}
CSDO_FN bool grow_box(double xc, double yc, const double* obs, int n_obs, double dimx, double dimy, double rv,
                      BoxD& res, const BoxCache& ec, const ObsMask& M) {
  const double ds = 0.1;
  // round in which a side's trial fails for a reason of its own: the map border, or the step after the last allowed one
  int stop0 = 1, stop1 = 1, stop2 = 1, stop3 = 1;
  {
    const double x_hi = dimx - rv, y_hi = dimy - rv;
    if (!(xc < rv || xc > x_hi || yc < rv || yc > y_hi)) {
      const int lim = GROW_LIMIT + 1;
      int v;
      v = first_step_of<0>(xc, yc, x_hi, y_hi, 0.0); stop0 = v < 1 ? 1 : (v > lim ? lim : v);
      v = first_step_of<1>(xc, yc, rv, y_hi, 0.0);   stop1 = v < 1 ? 1 : (v > lim ? lim : v);
      v = first_step_of<2>(xc, yc, x_hi, rv, 0.0);   stop2 = v < 1 ? 1 : (v > lim ? lim : v);
      v = first_step_of<3>(xc, yc, x_hi, y_hi, 0.0); stop3 = v < 1 ? 1 : (v > lim ? lim : v);
    }
  }
  int st0 = 0, st1 = 0, st2 = 0, st3 = 0;   // accepted steps of the sides that have stopped
  unsigned moving = 0xFu;
  bool seed_inside = false;
  int n_seen = 0;   // culled obstacles (counted by the first pass); the first min(n_seen, cap) have their word in the cache
#if defined(CSDO_ABL_BOX2X_PASSES)   // diagnostic: the passes twice, same results (what they cost = the difference)
  for (int rep_ = 0; rep_ < csdo_opaque_s(2); ++rep_) {
  st0 = st1 = st2 = st3 = 0; moving = 0xFu; seed_inside = false; n_seen = 0;
#endif
  for (int pass = 0; pass < 4 && !seed_inside; ++pass) {
    // earliest trial that fails, key = 4 * round + side
    int best = 4 * (GROW_LIMIT + 2);
    if (moving & 1u) best = 4 * stop0 + 0 < best ? 4 * stop0 + 0 : best;
    if (moving & 2u) best = 4 * stop1 + 1 < best ? 4 * stop1 + 1 : best;
    if (moving & 4u) best = 4 * stop2 + 2 < best ? 4 * stop2 + 2 : best;
    if (moving & 8u) best = 4 * stop3 + 3 < best ? 4 * stop3 + 3 : best;
    // An obstacle's four step counts E (one byte each, <= GROW_NEVER = 120) against the state of the pass, without a loop over
    // the sides.  The stopped sides must have their inequality true where they stopped (E <= accepted steps, <= 101): with the
    // bytes 0x80 | st (0x80 | 0x7f for a moving side) minus the bytes of E no byte borrows, and a byte keeps its top bit exactly
    // when E <= st.  Among the moving sides the obstacle is entered in round mx = max E by the LAST side that attains it:
    // 4 * mx + last is the maximum of 4 * E + side over the moving sides.
    const unsigned stp = 0x80808080u | (unsigned)((moving & 1u) ? 0x7f : st0) | ((unsigned)((moving & 2u) ? 0x7f : st1) << 8) |
                         ((unsigned)((moving & 4u) ? 0x7f : st2) << 16) | ((unsigned)((moving & 8u) ? 0x7f : st3) << 24);
    const unsigned emask = ((moving & 1u) ? 0xffu : 0u) | ((moving & 2u) ? 0xff00u : 0u) | ((moving & 4u) ? 0xff0000u : 0u) |
                           ((moving & 8u) ? 0xff000000u : 0u);
    const unsigned d1 = (moving & 2u) ? 1u : 0u, d2 = (moving & 4u) ? 2u : 0u, d3 = (moving & 8u) ? 3u : 0u;
    auto consider = [&](const unsigned packed) __attribute__((always_inline)) {
      const bool live = ((stp - packed) & 0x80808080u) == 0x80808080u;
      const unsigned pm = packed & emask;
      const unsigned k0 = (pm & 0xffu) << 2, k1 = (((pm >> 8) & 0xffu) << 2) | d1, k2 = (((pm >> 16) & 0xffu) << 2) | d2,
                     k3 = ((pm >> 24) << 2) | d3;
      const unsigned k01 = k0 > k1 ? k0 : k1, k23 = k2 > k3 ? k2 : k3;
      const int key = (int)(k01 > k23 ? k01 : k23);
      if (live && key < 4) seed_inside = true;   // every inequality holds at the seed: no trial can succeed
      if (live && key < best) best = key;
    };
    if (pass == 0) {
      ObsWalk w{M.m[0], M.m[1], M.m[2], M.m[3], OBS_MASK_CAP};
      int k;
      while (next_obstacle(w, n_obs, k)) {
        const unsigned packed = obstacle_steps(xc, yc, obs[k], obs[n_obs + k], obs[2 * n_obs + k]);   // (radius staged as r_obs + rv)
        if (n_seen < ec.cap) ec.base[(size_t)n_seen * (size_t)ec.stride] = packed;
        ++n_seen;
        consider(packed);
      }
    } else {
      const int n_cached = n_seen < ec.cap ? n_seen : ec.cap;
      for (int j = 0; j < n_cached; ++j) consider(ec.base[(size_t)j * (size_t)ec.stride]);
      if (n_seen > n_cached) {   // more obstacles in reach than cache slots: the rest is recomputed
        ObsWalk w{M.m[0], M.m[1], M.m[2], M.m[3], OBS_MASK_CAP};
        int k, j = 0;
        while (next_obstacle(w, n_obs, k)) {
          if (j++ < n_cached) continue;
          consider(obstacle_steps(xc, yc, obs[k], obs[n_obs + k], obs[2 * n_obs + k]));
        }
      }
    }
    const int side = best & 3, round = best >> 2;
    if (side == 0) st0 = round - 1;
    if (side == 1) st1 = round - 1;
    if (side == 2) st2 = round - 1;
    if (side == 3) st3 = round - 1;
    moving &= ~(1u << side);
  }
#if defined(CSDO_ABL_BOX2X_PASSES)
  }
#endif
  if (seed_inside) {   // (only ever found in the first pass: later, an obstacle that holds the box would have stopped a side)
    res = BoxD{xc, yc, xc, yc};
    return false;
  }
  // replay: the coordinates are the same sums as in the reference's walk
  BoxD box{xc, yc, xc, yc};
#if defined(CSDO_ABL_BOX2X_REPLAY)
  for (int rep_ = 0; rep_ < csdo_opaque_s(2); ++rep_) {
  box = BoxD{xc, yc, xc, yc};
#endif
  box.y_max = add_steps(yc, ds, st0 < GROW_LIMIT ? st0 : GROW_LIMIT);
  box.x_min = add_steps(xc, -ds, st1 < GROW_LIMIT ? st1 : GROW_LIMIT);
  box.y_min = add_steps(yc, -ds, st2 < GROW_LIMIT ? st2 : GROW_LIMIT);
  box.x_max = add_steps(xc, ds, st3 < GROW_LIMIT ? st3 : GROW_LIMIT);
#if defined(CSDO_ABL_BOX2X_REPLAY)
  }
#endif
  res = box;
  return (st0 + st1 + st2 + st3) > 0;
}

// generateBox, corridor.cc:124-159.  Returns bit0 = success, bits1-2 = initial status (0 legal, 1 out of map,
// 2 inside an inflated obstacle).  First colliding obstacle = lowest input index (documented deviation from the
// reference's unordered_set iteration order).
CSDO_FN int make_box(double x, double y, const double* obs, int n_obs, double dimx, double dimy, double rv,
                     BoxD& res, const BoxCache& ec = BoxCache{nullptr, 0, 0}) {
#if defined(CSDO_ABL_NOBOX)   // allocation experiment only
  res = BoxD{x - 1.0, y - 1.0, x + 1.0, y + 1.0};
  return 1;
#endif
  int initial = 0;
  if (x < rv || x > dimx - rv || y < rv || y > dimy - rv) {  // isPointOutOfMap + projectNearBorder
    initial = 1;
    const double eps = 1e-3, x0 = x, y0 = y;
    if (x0 < rv) x = rv + eps;
    else if (x0 > dimx - rv) x = dimx - rv - eps;
    if (y0 < rv) y = rv + eps;
    else if (y0 > dimy - rv) y = dimy - rv - eps;
  }
  int hit = -1;   // isPointCollision, and the obstacles within reach of a box grown from the point, in one walk
  ObsMask M = cull_obstacles(x, y, obs, n_obs, rv, hit);
#if defined(CSDO_ABL_BOX2X_CULL)
  for (int rep_ = 1; rep_ < csdo_opaque_s(2); ++rep_) M = cull_obstacles(x, y, obs + csdo_opaque_s(0), n_obs, rv, hit);
#endif
  // One call site for the growth: from the point itself, or (generateLegalPoint, corridor.cc:84-122) from up to 20 points on
  // a circle around the obstacle the point is inside of, alternating sides, until a grown box is valid against every obstacle.
  bool success = false;
  double hx = 0.0, hy = 0.0, theta0 = 0.0, d_ring = 0.0;
  if (hit >= 0) {
    initial = 2;
    hx = obs[hit];
    hy = obs[n_obs + hit];
    theta0 = atan2_of(y - hy, x - hx);
    d_ring = obs[2 * n_obs + hit] + 0.2;   // (rv + r_obs) + 0.2: the radius is staged as r_obs + rv
  }
  const int n_try = (hit >= 0) ? 20 : 1;
  BoxD cand{x, y, x, y};
  for (int i = 0; i < n_try && !success; ++i) {
    bool in_map = true;
    if (hit >= 0) {
      int j = i / 2;
      if (i % 2 == 1) j = -j;
      const double theta = theta0 + j * 2 * M_PI / 20;
      const SinCos sc_ = sincos_of(theta);
      const double sth = sc_.s, cth = sc_.c;
      x = hx + d_ring * cth;
      y = hy + d_ring * sth;
      in_map = x > rv && x < dimx - rv && y > rv && y < dimy - rv;
    }
    if (in_map) {
      if (hit >= 0) {   // (a point on the ring around the obstacle: its own neighbourhood)
        int unused_;
        M = cull_obstacles(x, y, obs, n_obs, rv, unused_);
      }
      const bool grew = grow_box(x, y, obs, n_obs, dimx, dimy, rv, cand, ec, M);
      if (hit >= 0) {
        ObsMask all;
        all.m[0] = all.m[1] = all.m[2] = all.m[3] = ~0ull;   // isBoxValid over every obstacle (bits >= n_obs masked below)
        if (n_obs < 256) {
          CSDO_FOR(w, 4, {
            const int lo_ = 64 * w;
            all.m[w] = (n_obs <= lo_) ? 0ull : ((n_obs - lo_ >= 64) ? ~0ull : ((1ull << (n_obs - lo_)) - 1ull));
          });
        }
        success = box_valid(cand, obs, n_obs, dimx, dimy, rv, all);
      } else {
        success = grew;
      }
    }
  }
  if (hit >= 0 && !success) cand = BoxD{x, y, x, y};  // zero-area fallback at the last point tried
  res = cand;
  return (success ? 1 : 0) | (initial << 1);
}

// A box at one call site of the agent program (inlined, four sites: as a REAL call - one copy of the largest cold piece, its LDS arrays
// rebuilt in the callee from their offsets - it was neutral on map100, -0.4 % on map50 and +2 % on room50 from the row role only, and
// put six scratch reloads into the solve from both roles: round 5).
CSDO_FN int box_at(const double x, const double y, const double* obs, const int n_obs, const double dimx, const double dimy,
                   const double rv, BoxD& res, const BoxCache& ec) {
#if defined(CSDO_ABL_BOX2X_ALL)   // diagnostic: every box twice
  for (int rep_ = 1; rep_ < csdo_opaque_s(2); ++rep_) make_box(x, y, obs + csdo_opaque_s(0), n_obs, dimx, dimy, rv, res, ec);
#endif
  return make_box(x, y, obs, n_obs, dimx, dimy, rv, res, ec);
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
  CSDO_FOR(r, 6, { out[r] = 0.0; });
  CSDO_FOR(c, 6, {  // column sweep: six independent accumulators
    CSDO_FOR(r, 6, { out[r] = fma(S[sym(r, c)], v[c], out[r]); });
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

CSDO_FN double rho_of_masks(unsigned eqmask, unsigned loosemask, int i, double rho) {
  const unsigned bit = 1u << i;
  return (loosemask & bit) ? RHO_MIN : ((eqmask & bit) ? RHO_EQ_OVER_RHO_INEQ * rho : rho);
}
CSDO_FN double rho_of(const LaneState& S, int i, double rho) { return rho_of_masks(S.eqmask, S.loosemask, i, rho); }

// Row classes known at compile time: kinematic and start/goal rows (0..6) have l = u, i.e. always OSQP's equality
// class; trust, control and steer rows (11..15) have u - l >= 0.1, always the inequality class; only the corridor rows
// (7..10) can be either (a zero-width box is an equality).  Home rows always have finite bounds (never "loose").
// Passing the two uniform values keeps rho_i out of per-lane registers.
template <int I>
CSDO_FN double rho_row(const LaneState& S, double rho, double rho_eq) {
  if constexpr (I < 7) return rho_eq;
  else if constexpr (I >= 11) return rho;
  else return (S.eqmask & (1u << I)) ? rho_eq : rho;
}

// MODE: where the iteration state of an ADMM block lives, chosen per agent by its working set (dsqp_kernel.hip).  The
// coupling F_l and two thirds of F_r of a BCR node (60 doubles) sit in its solver lane's REGISTERS in modes 0 and 1.
//   0  LDS: exchange vectors, bounds of the home rows, the rest of the factor (80 doubles per timestep); per agent
//      (AgentDesc::rows_lds, a run-time flag so that one kernel and one queue serve both kinds) also the duals / slacks of
//      the inter-vehicle rows and their rhs shares (13 doubles per plane) - otherwise those stay in the L2-resident
//      workspace
//   1  the third of the factor that is not in registers comes from the workspace too (horizons 235..256; 46 per timestep)
//   3  horizons beyond 256 (1024 threads, 128 registers per lane): factor, bounds and rows from the workspace; LDS only
//      holds the exchange vectors (30 doubles per timestep)
template <int ROLE, int MODE, class RowStore, class SolvStore>
CSDO_FN void agent_program(const DeviceBatch& B, const int agent, const Shm& sh, RowStore&& lanes_r,
                           SolvStore&& lanes_s, ProgramOut& out);

}  // namespace csdo

#include "dsqp_program_impl.h"
