// dsqp_kernel_body.h — the per-agent kernel template and its launcher.  Each instantiation is compiled in its own
// translation unit (dsqp_variant.hip with -DCSDO_V_BLOCK/-DCSDO_V_MODE/-DCSDO_V_SPLIT) so the build parallelises.
#pragma once
#define CSDO_LANE_MODE_DEVICE 1
#include "dsqp_program.h"
#include "dsqp_launch.h"

namespace csdo {

// BLOCK = 2 * (lanes per role): threads [0, BLOCK/2) are row lanes, [BLOCK/2, BLOCK) solver lanes (dsqp_program.h)
// MODE: where an ADMM block keeps its iteration state (0: LDS + registers, 1: the factor's LDS part from the workspace,
// 3: long horizons), see agent_program in dsqp_program.h
// SPLIT: two specialised lanes per timestep (row waves + solver waves)
// LDS carve of one agent (see Shm); `lds` is the workgroup's dynamic LDS.  Every array starts 16-byte aligned (even
// strides) and says so, which lets the compiler use ds_read_b128 / ds_write_b128 on the lane-major arrays.
template <class T>
__device__ __forceinline__ T* aligned16(T* p) {
  __builtin_assume(((unsigned long long)p & 15ull) == 0ull);
  return (T*)__builtin_assume_aligned(p, 16);
}

template <int MODE, bool BIGT>
__device__ __forceinline__ Shm carve_lds(const DeviceBatch& B, const int agent, double* lds, const int lds_doubles) {
  const int ad_Nt = uniform_i32(B.agents[agent].Nt);
  const long long ad_fac_off = uniform_i64(B.agents[agent].fac_off);
  const long long ad_rows_off = uniform_i64(B.agents[agent].rows_off);
  const int ad_n_planes = uniform_i32(B.agents[agent].n_planes);
  const int n_obs = uniform_i32(B.worlds[uniform_i32(B.agents[agent].world)].n_obs);
  // the stride is built as 2 * (half): every array offset below is then a visible multiple of 16 bytes, and the
  // compiler may use 16-byte LDS accesses on the lane-major arrays (with a runtime "& ~1" it cannot prove that and
  // falls back to ds_read2_b64, twice the LDS cycles)
  const int st = 2 * ((ad_Nt + 1) >> 1);
  Shm sh;
  sh.stride = st;
  sh.wave0 = uniform_i32((int)threadIdx.x & ~63);
  sh.vec = aligned16(lds);
  sh.pl = sh.vec;       // aliases, see Shm
  sh.pr = aligned16(sh.vec + 6 * st);
  // (mode 2: the rhs shares of the row lanes wait in the right partials' array: written in the update, behind the solve's last
  //  barrier, read by the timestep's solver lane before its wave writes any partial of the next solve, all writers of pr[t] being in
  //  the wave of lane t)
  sh.rhs = MODE == 2 ? sh.pr : aligned16(sh.pr + 6 * st);
  sh.carry = aligned16(sh.rhs + 6 * st);
  sh.red = sh.vec;
  double* rest = sh.carry + 6 * st;
  if constexpr (MODE == 2) {   // the lean layout plus the LDS part of the lane's second block; carry2 is only used between blocks
    sh.stash = sh.vec;
    sh.lohi = nullptr;
    sh.fx = aligned16(rest);
    sh.carry2 = sh.fx;
    rest = sh.fx + LD_fx2 * st;
  } else if constexpr (MODE != 3) {
    sh.stash = sh.vec;
    sh.lohi = aligned16(rest);
    sh.carry2 = sh.lohi;
    rest = sh.lohi + 22 * st;
    sh.fx = aligned16(rest);
    if constexpr (MODE == 0) rest = sh.vec + LD_block * st;
    else rest = sh.fx + LD_fx1 * st;   // mode 1: the LDS part of the lane's second block only
  } else {
    sh.stash = sh.lohi = sh.fx = nullptr;
    sh.carry2 = aligned16(rest);
    rest = sh.carry2 + 6 * st;
  }
  const int n_obs_pad = 2 * ((3 * n_obs + 1) >> 1);
  sh.obs = rest;
  sh.bcast = aligned16(sh.obs + n_obs_pad);
  sh.tvec = aligned16(sh.bcast + 32);
  if constexpr (BIGT) {   // the tail's capacity is the agent's (dsqp_class.h; the same sizes as dsqp_lds_bytes)
    const int tn = uniform_i32(B.agents[agent].tail_nodes);
    const int cap = tn > TAIL_NODES ? 6 * tn : TAIL_N;
    sh.tvec_half = tn > TAIL_NODES ? TAIL_N_BIG : TAIL_N;
    sh.ld_tinv = cap + 2;
    sh.tinv = aligned16(sh.tvec + 2 * sh.tvec_half);
    rest = sh.tinv + cap * (cap + 2);
  } else {
    sh.tvec_half = TAIL_N;
    sh.ld_tinv = 38;
    sh.tinv = aligned16(sh.tvec + 2 * TAIL_N);
    rest = sh.tinv + TAIL_N * 38;
  }
  sh.pcg = B.rows_ws + ad_rows_off * ROWS_WS_STRIDE + (size_t)32 * ad_n_planes;
  if constexpr (MODE == 0) {   // only used by agents with AgentDesc::rows_lds (the launch sized the LDS for them)
    const int n_pc_pad = 2 * ((3 * ad_n_planes + 1) >> 1);
    sh.pc = aligned16(rest);
    sh.prow = aligned16(sh.pc + n_pc_pad);
    sh.pco = aligned16(sh.prow + (size_t)LD_prow * ad_n_planes);
    sh.n_pco = sh.n_pco_ld = 0;
    if (uniform_i32(B.agents[agent].rows_lds) != 0) {   // (the launch sized the LDS for pc / prow of such agents)
      const int left = lds_doubles - (int)(sh.pco - lds) - 2;
      int fit = left > 0 ? left / 16 : 0;
      fit &= ~1;
      sh.n_pco = fit < ad_n_planes ? fit : ad_n_planes;
      sh.n_pco_ld = 2 * ((sh.n_pco + 1) >> 1);
    }
  } else {
    sh.pc = sh.prow = sh.pco = nullptr;
    sh.n_pco = sh.n_pco_ld = 0;
  }
  double* fac_global = B.fac_ws + ad_fac_off;
  sh.facE = fac_global;
  sh.facX = fac_global + (size_t)FAC_E_DOUBLES * st;
  sh.cold = sh.facX + (size_t)FAC_X_DOUBLES * st;
  return sh;
}

// Persistent workgroups: a launch has at most one workgroup per CU and each takes agents off the group's queue (the
// launch order of capi.hip: heaviest first) until it is empty.  Workgroups of a plain grid are dealt round-robin to the
// 8 XCDs, every XCD refills only its own CUs in order, and a finished CU waited 1.8 ms on average (0.5 ms median) for
// its next workgroup: 14 % of the CU time of a 3000-agent batch.  Each role runs its own loop (same barriers in both).
// The 256-thread class (horizons <= 128) is built for two workgroups per CU: a workgroup's waves spend most of a step waiting
// for LDS round trips and barriers (scripts/microbench2.hip), a second agent on the same SIMDs fills those gaps; its working
// set (<= 80 KB of LDS, 256 registers per lane) allows it.  The 512-thread class fills the register file by itself.
#if !defined(CSDO_PRIO_SOLVER)
#define CSDO_PRIO_SOLVER 2
#endif
template <int BLOCK, int MODE, bool SPLIT, int REFINE = 0>
__global__ __launch_bounds__(BLOCK, (BLOCK == 256 ? 2 : 1)) void dsqp_agent_kernel(const DeviceBatch B, const int first, const int count,
                                                             int* __restrict__ queue, const int lds_doubles) {
  extern __shared__ __align__(16) double lds[];
  __shared__ int next_in_queue;
  static_assert(SPLIT, "one thread per timestep playing both roles is only built lane-serially (tests/emu)");
  if (threadIdx.x < BLOCK / 2) {        // row waves
    for (;;) {
      if (threadIdx.x == 0) next_in_queue = atomicAdd(queue, 1);
      __syncthreads();
      const int q_idx = uniform_i32(next_in_queue);
      __syncthreads();
      if (q_idx >= count) break;
      const int agent = uniform_i32(B.order[first + q_idx]);
      const long long t_begin = wall_clock64();
#if defined(CSDO_PROFILE_PHASES)
      if (threadIdx.x == 0 && B.prof) {   // diagnostic: start time (100 MHz ticks) and where the workgroup runs
        B.prof[(int64_t)agent * 48 + 47] = t_begin;
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (31 << 11));    // HW_REG_XCC_ID
        B.prof[(int64_t)agent * 48 + 46] = ((long long)(xcc & 0xf) << 32) | hw;
      }
#endif
      const Shm sh = carve_lds<MODE, (BLOCK == 512)>(B, agent, lds, lds_doubles);
      ProgramOut po;
      RowRegs lr;
      SolvRegs ls_unused;
      agent_program<ROLE_ROW, MODE, (BLOCK == 512), REFINE>(B, agent, sh, lr, ls_unused, po);
      if (threadIdx.x == 0) {
        B.sqp_iters[agent] = po.sqp_iters;
        B.admm_iters[agent] = po.admm_iters;
        B.last_status[agent] = po.last_status;
        B.static_legal[agent] = po.static_legal;
        B.agent_ticks[agent] = wall_clock64() - t_begin;
      }
    }
  } else {                              // solver waves
    // The solver waves' instructions go first wherever a SIMD's two waves - one of each role - both have something to issue: every
    // ADMM iteration's critical path runs through the solver waves, the row waves' work beside it (tail product, prefetches) has
    // slack.  One s_setprio per workgroup (levels 1, 2, 3 measure the same; the row waves raised instead: no change): map100 53.9 ->
    // 53.3 ms, room50 36.7 -> 36.2, agents100 47.3 -> 46.3, one 50-agent instance alone 8.85 -> 8.3 ms.  Same bits.
    __builtin_amdgcn_s_setprio(CSDO_PRIO_SOLVER);
    for (;;) {
      __syncthreads();
      const int q_idx = uniform_i32(next_in_queue);
      __syncthreads();
      if (q_idx >= count) break;
      const int agent = uniform_i32(B.order[first + q_idx]);
      const Shm sh = carve_lds<MODE, (BLOCK == 512)>(B, agent, lds, lds_doubles);
      ProgramOut po;
      RowRegs lr_unused;
      SolvRegs ls;
      agent_program<ROLE_SOLVER, MODE, (BLOCK == 512), REFINE>(B, agent, sh, lr_unused, ls, po);
    }
  }
}

template <int BLOCK, int MODE, bool SPLIT, int REFINE>
hipError_t launch_variant(const DeviceBatch& B, const LaunchGroup& g, int workgroups, hipStream_t stream) {
  auto kernel = dsqp_agent_kernel<BLOCK, MODE, SPLIT, REFINE>;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, dim3(workgroups), dim3(BLOCK), g.lds_bytes, stream, B, g.first, g.count, g.queue,
                     (int)(g.lds_bytes / sizeof(double)));
  return hipGetLastError();
}

}  // namespace csdo
