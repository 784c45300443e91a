// dsqp_kernel.hip — gfx950 kernels of the DO backend.
//   dsqp_agent_kernel<BLOCK,BIG,SPLIT> (dsqp_kernel_body.h, one translation unit per instantiation): one workgroup per agent runs the whole per-agent SQP (dsqp_program.h) start to
//     finish; grid = number of agents in the batch; no host round trips, no inter-workgroup communication
//     (agents are independent once the separating planes are fixed, sqp/dsqp_solver.cc:1198-1205).
//   box_kernel: one lane per point, safe boxes for arbitrary points (csdo_generate_boxes).
#define CSDO_LANE_MODE_DEVICE 1
#include "dsqp_program.h"
#include "dsqp_launch.h"

#include <cstdlib>

namespace csdo {

template <int BLOCK, int MODE, bool SPLIT>
hipError_t launch_variant(const DeviceBatch& B, const LaunchGroup& g, int workgroups, hipStream_t stream);   // dsqp_variant.hip

__global__ void box_kernel(const double* __restrict__ pts, int n, const double* __restrict__ obs_aos, int n_obs,
                           double dimx, double dimy, double rv, double* __restrict__ boxes,
                           int* __restrict__ status) {
  extern __shared__ __align__(16) double lds[];
  for (int k = (int)threadIdx.x; k < n_obs; k += (int)blockDim.x) {
    lds[k] = obs_aos[3 * k];
    lds[n_obs + k] = obs_aos[3 * k + 1];
    lds[2 * n_obs + k] = obs_aos[3 * k + 2] + rv;   // (make_box takes the radii inflated by the disc radius)
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

// the program's sin / cos / tan / atan2 (csdo_math.h) on arbitrary arguments: what csdo_math_eval runs (a diagnostic entry - the
// parity tests compare its bits with the host build of the same header)
__global__ void math_probe_kernel(int fn, const double* __restrict__ a, const double* __restrict__ b, double* __restrict__ out, int n) {
  const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (i >= n) return;
  double r;
  if (fn == 0) r = sincos_of(a[i]).s;
  else if (fn == 1) r = sincos_of(a[i]).c;
  else if (fn == 2) r = tan_of(a[i]);
  else r = atan2_of(a[i], b[i]);
  out[i] = r;
}
hipError_t launch_math_probe(int fn, const double* a, const double* b, double* out, int n, hipStream_t stream) {
  hipLaunchKernelGGL(math_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fn, a, b, out, n);
  return hipGetLastError();
}

size_t dsqp_lds_bytes(int nt, int n_obs, int n_planes, int mode, bool rows_lds) {
  const int st = (nt + 1) & ~1;
  // exchange vectors vec, pr, rhs, carry (6 each) + bounds of the home rows 22 (the t -> t-1 hand-over aliases them) +
  // the third of the factor that is not in the solver lane's registers 34 (mode 0); mode 3: vec, pr, rhs, carry, carry2
  const size_t per_lane = mode == 3 ? 30 : (mode == 2 ? (size_t)LD_block2 : (mode == 1 ? (size_t)LD_block1 : (size_t)LD_block));
  const size_t n_obs_pad = (3 * (size_t)n_obs + 1) & ~(size_t)1, n_pc_pad = (3 * (size_t)n_planes + 1) & ~(size_t)1;
  const size_t planes = (mode == 0 && rows_lds) ? n_pc_pad + (size_t)LD_prow * n_planes : 0;   // rhs shares + duals / slacks
  return (per_lane * st + n_obs_pad + 32 + 2 * TAIL_N + TAIL_N * 38 + planes) * sizeof(double);
}

constexpr size_t LDS_CAP = 160 * 1024 - 64;   // 160 KB per workgroup minus the kernel's static LDS (queue slot)
constexpr size_t LDS_CAP_2WG = 80 * 1024 - 64;   // two workgroups of the 256-thread class per CU
size_t dsqp_lds_capacity() { return LDS_CAP; }
size_t dsqp_lds_capacity_two_per_cu() { return LDS_CAP_2WG; }
int dsqp_workgroups_per_cu(int block, size_t lds_bytes) { return (block == 256 && lds_bytes <= LDS_CAP_2WG) ? 2 : 1; }

int dsqp_agent_class(int nt, int n_obs, int n_planes, int* mode, int* rows_lds) {
  // workgroup size: two specialised lanes per timestep (Nt <= 128: 256 threads, <= 256: 512 threads, <= 512: 1024
  // threads with 128 registers per lane: correct but spills; horizons that long are outside the benchmark sets)
  // The 256-thread class runs two workgroups per CU, so it only takes agents whose working set fits half the LDS; a
  // short horizon that does not (Nt > ~105 with 25 obstacles) runs in the 512-thread class with half its lanes idle.
  // Horizons 257 .. 384 take 768 threads: three waves per SIMD leave 168 registers per lane instead of 128 (measured on the
  // room set, whose long agents have 257 .. 295 timesteps).
  int block = nt <= 128 ? 256 : (nt <= 256 ? 512 : (nt <= 384 ? 768 : 1024));
  if (block == 256 && dsqp_lds_bytes(nt, n_obs, n_planes, 0, false) > LDS_CAP_2WG) block = 512;
  *rows_lds = 0;
  if (block >= 768) {   // nothing of the factor in registers: F_r in LDS where that fits (mode 2), else from the workspace
    *mode = (block == 768 && dsqp_lds_bytes(nt, n_obs, n_planes, 2, false) <= LDS_CAP) ? 2 : 3;
  } else if (dsqp_lds_bytes(nt, n_obs, n_planes, 0, true) <= (block == 256 ? LDS_CAP_2WG : LDS_CAP)) {   // (256: keep two per CU)
    *mode = 0;
    *rows_lds = 1;
  } else {
    *mode = dsqp_lds_bytes(nt, n_obs, n_planes, 0, false) <= LDS_CAP ? 0 : 1;
    // An obstacle list that does not even fit beside the lean 512-thread layout (mode 1: 52 doubles per timestep) runs in the
    // 768-thread class, whose modes keep 52 / 30 doubles per timestep in LDS - with lanes to spare for a horizon this short,
    // slower, but it runs: 5000 obstacles beside 100 timesteps, 3900 beside 200 (mode 1's growth from 46 to 52
    // doubles had turned worlds away that round 3 accepted; the reference has no such limit at all).
    if (*mode == 1 && dsqp_lds_bytes(nt, n_obs, n_planes, 1, false) > LDS_CAP) {
      block = 768;
      *mode = dsqp_lds_bytes(nt, n_obs, n_planes, 2, false) <= LDS_CAP ? 2 : 3;
    }
  }
  return block;
}

hipError_t launch_dsqp(const DeviceBatch& B, const LaunchGroup& g, int workgroups, hipStream_t stream) {
  if (g.count <= 0 || workgroups <= 0) return hipSuccess;
  if (g.lds_bytes > LDS_CAP) return hipErrorInvalidValue;
  switch (g.block * 10 + g.mode) {
    case 2560: return launch_variant<256, 0, true>(B, g, workgroups, stream);
    case 5120: return launch_variant<512, 0, true>(B, g, workgroups, stream);
    case 5121: return launch_variant<512, 1, true>(B, g, workgroups, stream);
    case 7682: return launch_variant<768, 2, true>(B, g, workgroups, stream);
    case 7683: return launch_variant<768, 3, true>(B, g, workgroups, stream);
    case 10243: return launch_variant<1024, 3, true>(B, g, workgroups, stream);
  }
  return hipErrorInvalidValue;
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
