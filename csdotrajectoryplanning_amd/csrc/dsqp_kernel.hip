// dsqp_kernel.hip — gfx950 kernels of the DO backend.
//   dsqp_agent_kernel<BLOCK>: one workgroup per agent runs the whole per-agent SQP (dsqp_program.h) start to
//     finish; grid = number of agents in the batch; no host round trips, no inter-workgroup communication
//     (agents are independent once the separating planes are fixed, sqp/dsqp_solver.cc:1198-1205).
//   box_kernel: one lane per point, safe boxes for arbitrary points (csdo_generate_boxes).
#define CSDO_LANE_MODE_DEVICE 1
#include "dsqp_program.h"
#include "dsqp_launch.h"

#include <cstdlib>

namespace csdo {

// BLOCK = 2 * (lanes per role): threads [0, BLOCK/2) are row lanes, [BLOCK/2, BLOCK) solver lanes (dsqp_program.h)
// BIG: E_r and the bounds stay in the workspace and LDS holds only the 6-vectors (horizons / obstacle counts whose
// working set exceeds 160 KB of LDS)
// SPLIT: two specialised lanes per timestep (row waves + solver waves); otherwise one thread per timestep does both
template <int BLOCK, bool BIG, bool SPLIT>
__global__ __launch_bounds__(BLOCK) void dsqp_agent_kernel(const DeviceBatch B, const int max_obs, const int max_planes) {
  extern __shared__ __align__(16) double lds[];
  const int agent = (int)blockIdx.x;
  if (agent >= B.n_agents) return;
  const long long t_begin = wall_clock64();
  const int ad_Nt = uniform_i32(B.agents[agent].Nt);
  const long long ad_fac_off = uniform_i64(B.agents[agent].fac_off);
  const long long ad_rows_off = uniform_i64(B.agents[agent].rows_off);
  const int ad_n_planes = uniform_i32(B.agents[agent].n_planes);
  const int st = (ad_Nt + 1) & ~1;
  Shm sh;
  sh.stride = st;
  sh.vec = lds;
  sh.pl = sh.vec + 6 * st;
  sh.pr = sh.pl + 6 * st;
  sh.carry = sh.pl;     // aliases, see Shm
  sh.carry2 = sh.pr;
  sh.lohi = sh.pr + 6 * st;
  sh.red = sh.lohi;                       // BIG: this region is only the 12-wide reduction scratch
  sh.sinvs = sh.lohi + 22 * st;
  sh.er = sh.sinvs + 22 * st;
  sh.obs = BIG ? (sh.red + 12 * st) : (sh.er + 38 * st);
  sh.bcast = sh.obs + 3 * max_obs;
  sh.tvec = sh.bcast + 32;
  sh.tinv = sh.tvec + 2 * TAIL_N;
  sh.pc = BIG ? (B.rows_ws + ad_rows_off * ROWS_WS_STRIDE + (size_t)32 * ad_n_planes) : (sh.tinv + TAIL_N * 38);
  (void)max_planes;
  double* fac_global = B.fac_ws + ad_fac_off;
  sh.facE = fac_global;
  sh.facX = fac_global + (size_t)FAC_E_DOUBLES * st;
  sh.cold = sh.facX + (size_t)FAC_X_DOUBLES * st;
  ProgramOut po;
  if constexpr (!SPLIT) {
    RowRegs lr;
    SolvRegs ls;
    agent_program<ROLE_BOTH, BIG>(B, agent, sh, lr, ls, po);
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
    agent_program<ROLE_ROW, BIG>(B, agent, sh, lr, ls_unused, po);
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
    agent_program<ROLE_SOLVER, BIG>(B, agent, sh, lr_unused, ls, po);
  }
}

__global__ void box_kernel(const double* __restrict__ pts, int n, const double* __restrict__ obs_aos, int n_obs,
                           double dimx, double dimy, double rv, double* __restrict__ boxes,
                           int* __restrict__ status) {
  extern __shared__ __align__(16) double lds[];
  for (int k = (int)threadIdx.x; k < n_obs; k += (int)blockDim.x) {
    lds[k] = obs_aos[3 * k];
    lds[n_obs + k] = obs_aos[3 * k + 1];
    lds[2 * n_obs + k] = obs_aos[3 * k + 2];
  }
  __syncthreads();
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  BoxD b{0, 0, 0, 0};
  status[i] = make_box(pts[2 * i], pts[2 * i + 1], lds, n_obs, dimx, dimy, rv, b);
  boxes[4 * i] = b.x_min;
  boxes[4 * i + 1] = b.y_min;
  boxes[4 * i + 2] = b.x_max;
  boxes[4 * i + 3] = b.y_max;
}

size_t dsqp_lds_bytes(int max_nt, int max_obs, int max_planes, bool big) {
  const int st = (max_nt + 1) & ~1;
  const size_t per_lane = big ? 30 : 100;  // vec 6 + pl 6 + pr 6 + (red 12 | lohi 22 + sinv 22 + er 38)
  return (per_lane * st + (size_t)3 * max_obs + 32 + 2 * TAIL_N + TAIL_N * 38 + (big ? 0 : (size_t)3 * max_planes)) *
         sizeof(double);
}

hipError_t launch_dsqp(const DeviceBatch& B, int max_nt, int max_obs, int max_planes, hipStream_t stream) {
  constexpr size_t LDS_CAP = 160 * 1024;
  DeviceBatch b = B;
  const bool big = max_nt > 256 || dsqp_lds_bytes(max_nt, max_obs, max_planes, false) > LDS_CAP;   // non-BIG fits up to Nt ~ 200
  b.lds_fac = big ? 0 : 1;
  const size_t bytes = dsqp_lds_bytes(max_nt, max_obs, max_planes, big);
  if (bytes > LDS_CAP) return hipErrorInvalidValue;
  auto go = [&](auto kernel, int block) -> hipError_t {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(b.n_agents), dim3(block), bytes, stream, b, max_obs, max_planes);
    return hipGetLastError();
  };
  // Default: two specialised lanes per timestep (Nt <= 128: 256 threads, <= 256: 512 threads, <= 512: 1024 threads with
  // 128 registers per lane: correct but spills; horizons that long are outside the benchmark sets).
  // CSDO_SINGLE_ROLE=1 selects one thread per timestep playing both roles (256 threads, 512 registers per lane) for
  // Nt <= 256: measured 10 % slower (no overlap of the inter-vehicle pass with the row update); kept for comparison.
  static const bool single = [] { const char* e = getenv("CSDO_SINGLE_ROLE"); return e && e[0] == '1'; }();
  if (max_nt <= 256 && single)
    return big ? go(dsqp_agent_kernel<256, true, false>, 256) : go(dsqp_agent_kernel<256, false, false>, 256);
  if (max_nt <= 128) return big ? go(dsqp_agent_kernel<256, true, true>, 256) : go(dsqp_agent_kernel<256, false, true>, 256);
  if (max_nt <= 256) return big ? go(dsqp_agent_kernel<512, true, true>, 512) : go(dsqp_agent_kernel<512, false, true>, 512);
  return go(dsqp_agent_kernel<1024, true, true>, 1024);
}

hipError_t launch_boxes(const double* pts, int n, const double* obs, int n_obs, double dimx, double dimy, double rv,
                        double* boxes, int* status, hipStream_t stream) {
  const int block = 64;
  const int grid = (n + block - 1) / block;
  hipLaunchKernelGGL(box_kernel, dim3(grid), dim3(block), (size_t)3 * n_obs * sizeof(double), stream, pts, n, obs,
                     n_obs, dimx, dimy, rv, boxes, status);
  return hipGetLastError();
}

}  // namespace csdo
