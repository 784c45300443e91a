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

template <int BLOCK, int MODE, bool SPLIT, int REFINE>
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

size_t dsqp_lds_capacity() { return LDS_CAP; }
size_t dsqp_lds_capacity_two_per_cu() { return LDS_CAP_2WG; }
int dsqp_workgroups_per_cu(int block, size_t lds_bytes) { return (block == 256 && lds_bytes <= LDS_CAP_2WG) ? 2 : 1; }

hipError_t launch_dsqp(const DeviceBatch& B, const LaunchGroup& g, int workgroups, hipStream_t stream) {
  if (g.count <= 0 || workgroups <= 0) return hipSuccess;
  if (g.lds_bytes > LDS_CAP) return hipErrorInvalidValue;
  const int key = g.block * 10 + g.mode;
#define CSDO_LAUNCH_TABLE(R)                                                                   \
  switch (key) {                                                                               \
    case 2560: return launch_variant<256, 0, true, R>(B, g, workgroups, stream);               \
    case 5120: return launch_variant<512, 0, true, R>(B, g, workgroups, stream);               \
    case 5121: return launch_variant<512, 1, true, R>(B, g, workgroups, stream);               \
    case 7682: return launch_variant<768, 2, true, R>(B, g, workgroups, stream);               \
    case 7683: return launch_variant<768, 3, true, R>(B, g, workgroups, stream);               \
    case 10243: return launch_variant<1024, 3, true, R>(B, g, workgroups, stream);             \
  }                                                                                            \
  return hipErrorInvalidValue
  // csdo_qp_parm::solve_refinement picks the instantiations with that refinement compiled in (the default ones contain none of it)
  if (B.prm.solve_refinement == 1) { CSDO_LAUNCH_TABLE(1); }
  if (B.prm.solve_refinement == 2) { CSDO_LAUNCH_TABLE(2); }
  CSDO_LAUNCH_TABLE(0);
#undef CSDO_LAUNCH_TABLE
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
