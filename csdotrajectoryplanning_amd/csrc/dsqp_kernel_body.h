// dsqp_kernel_body.h — the per-agent kernel template and its launcher.  Each instantiation is compiled in its own
// translation unit (dsqp_variant.hip with -DCSDO_V_BLOCK/-DCSDO_V_MODE/-DCSDO_V_SPLIT) so the build parallelises.
#pragma once
#define CSDO_LANE_MODE_DEVICE 1
#include "dsqp_program.h"
#include "dsqp_launch.h"

namespace csdo {

// BLOCK = 2 * (lanes per role): threads [0, BLOCK/2) are row lanes, [BLOCK/2, BLOCK) solver lanes (dsqp_program.h)
// MODE: LDS residency of an ADMM block (0 everything, 1 without pivot inverses and bounds, 2 only the 6-vectors),
// see agent_program in dsqp_program.h
// SPLIT: two specialised lanes per timestep (row waves + solver waves); otherwise one thread per timestep does both
template <int BLOCK, int MODE, bool SPLIT>
__global__ __launch_bounds__(BLOCK) void dsqp_agent_kernel(const DeviceBatch B, const int first, const int count) {
  extern __shared__ __align__(16) double lds[];
  if ((int)blockIdx.x >= count) return;
  const int agent = uniform_i32(B.order[first + (int)blockIdx.x]);
  const long long t_begin = wall_clock64();
  const int ad_Nt = uniform_i32(B.agents[agent].Nt);
  const long long ad_fac_off = uniform_i64(B.agents[agent].fac_off);
  const long long ad_rows_off = uniform_i64(B.agents[agent].rows_off);
  const int ad_n_planes = uniform_i32(B.agents[agent].n_planes);
  const int n_obs = uniform_i32(B.worlds[uniform_i32(B.agents[agent].world)].n_obs);
  const int st = (ad_Nt + 1) & ~1;
  Shm sh;
  sh.stride = st;
  sh.vec = lds;
  sh.pl = sh.vec;       // aliases, see Shm
  sh.pr = sh.vec + 6 * st;
  sh.carry = sh.pr;
  double* rest = sh.pr + 6 * st;
  if constexpr (MODE == 0) {
    sh.lohi = rest;
    sh.red = sh.lohi;                     // reductions only run between ADMM blocks
    sh.sinvs = sh.lohi + 22 * st;
    sh.er = sh.sinvs + 22 * st;
    sh.carry2 = sh.er;
    rest = sh.er + 38 * st;
  } else if constexpr (MODE == 1) {
    sh.lohi = sh.sinvs = nullptr;
    sh.er = rest;
    sh.red = sh.er;
    sh.carry2 = sh.er + 12 * st;
    rest = sh.er + 38 * st;
  } else {
    sh.lohi = sh.sinvs = sh.er = nullptr;
    sh.carry2 = rest;
    sh.red = sh.carry2 + 6 * st;
    rest = sh.red + 12 * st;
  }
  sh.obs = rest;
  sh.bcast = sh.obs + 3 * n_obs;
  sh.tvec = sh.bcast + 32;
  sh.tinv = sh.tvec + 2 * TAIL_N;
  sh.pc = MODE == 2 ? (B.rows_ws + ad_rows_off * ROWS_WS_STRIDE + (size_t)32 * ad_n_planes) : (sh.tinv + TAIL_N * 38);
  double* fac_global = B.fac_ws + ad_fac_off;
  sh.facE = fac_global;
  sh.facX = fac_global + (size_t)FAC_E_DOUBLES * st;
  sh.cold = sh.facX + (size_t)FAC_X_DOUBLES * st;
  ProgramOut po;
  if constexpr (!SPLIT) {
    RowRegs lr;
    SolvRegs ls;
    agent_program<ROLE_BOTH, MODE>(B, agent, sh, lr, ls, po);
    if (threadIdx.x == 0) {
      B.sqp_iters[agent] = po.sqp_iters;
      B.admm_iters[agent] = po.admm_iters;
      B.last_status[agent] = po.last_status;
      B.static_legal[agent] = po.static_legal;
      B.agent_ticks[agent] = wall_clock64() - t_begin;
    }
  } else if (threadIdx.x < BLOCK / 2) {        // row waves
    RowRegs lr;
    SolvRegs ls_unused;
    agent_program<ROLE_ROW, MODE>(B, agent, sh, lr, ls_unused, po);
    if (threadIdx.x == 0) {
      B.sqp_iters[agent] = po.sqp_iters;
      B.admm_iters[agent] = po.admm_iters;
      B.last_status[agent] = po.last_status;
      B.static_legal[agent] = po.static_legal;
      B.agent_ticks[agent] = wall_clock64() - t_begin;
    }
  } else {                              // solver waves
    RowRegs lr_unused;
    SolvRegs ls;
    agent_program<ROLE_SOLVER, MODE>(B, agent, sh, lr_unused, ls, po);
  }
}

template <int BLOCK, int MODE, bool SPLIT>
hipError_t launch_variant(const DeviceBatch& B, const LaunchGroup& g, hipStream_t stream) {
  auto kernel = dsqp_agent_kernel<BLOCK, MODE, SPLIT>;
  hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)g.lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kernel, dim3(g.count), dim3(BLOCK), g.lds_bytes, stream, B, g.first, g.count);
  return hipGetLastError();
}

}  // namespace csdo
