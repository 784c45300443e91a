// Calibration of one block-cyclic-reduction step on gfx950 (diagnostic, not part of the library): what a barrier-delimited
// step of the solver waves costs as a function of its content.  512 threads (4 idle "row" waves at the barrier + 4 solver
// waves), one workgroup per CU on `grid` CUs.   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off microbench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
template <int VARIANT>
__global__ __launch_bounds__(512) void step_kernel(double* out, long long* ticks, int rounds, int stride_h) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  for (int k = tid; k < 16384; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
  double m[36], a[6] = {1, 2, 3, 4, 5, 6};
#pragma unroll
  for (int k = 0; k < 36; ++k) m[k] = 1.0 + 1e-7 * (k + tid);
  __syncthreads();
  const bool solver = tid >= 256;
  const int t = tid - 256;
  const bool active = solver && ((t & (2 * stride_h - 1)) == stride_h);
  double* vec = lds;            // [256][6]
  double* pr = lds + 6 * 256;   // [256][6]
  double* big = lds + 12 * 256; // [256][34]
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    if (VARIANT == 0) {                 // empty step: barrier only
    } else if (VARIANT == 1) {          // LDS exchange only: read 12 doubles of neighbours, write 12
      if (active) {
        double b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = a[k] - pr[(t - stride_h) * 6 + k] - vec[((t + stride_h) & 255) * 6 + k];
#pragma unroll
        for (int k = 0; k < 6; ++k) { vec[t * 6 + k] = b[k]; pr[t * 6 + k] = b[k] * 0.5; a[k] = b[k]; }
      }
    } else if (VARIANT == 2 || VARIANT == 3) {   // + 108 (2) or 36 (3) multiply-adds on register matrices, 12 chains
      if (active) {
        double b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = a[k] - pr[(t - stride_h) * 6 + k] - vec[((t + stride_h) & 255) * 6 + k];
        double acc[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] = 0.0;
        constexpr int REP = VARIANT == 2 ? 3 : 1;
#pragma unroll
        for (int rep = 0; rep < REP; ++rep)
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              acc[c] = fma(m[rr * 6 + c], b[rr], acc[c]);
              acc[6 + c] = fma(m[(rr + 3) * 6 + c], b[rr + 3], acc[6 + c]);
            }
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double v = acc[k] + acc[6 + k]; vec[t * 6 + k] = v; pr[t * 6 + k] = v * 0.5; a[k] = v * 1e-3 + 1.0; }
      }
    } else if (VARIANT == 4) {          // + 34 doubles of the lane's own factor read from LDS (17 b128), 108 multiply-adds
      if (active) {
        double b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = a[k] - pr[(t - stride_h) * 6 + k] - vec[((t + stride_h) & 255) * 6 + k];
        double f[34];
#pragma unroll
        for (int k = 0; k < 34; ++k) f[k] = big[t * 34 + k];
        double acc[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] = 0.0;
#pragma unroll
        for (int rep = 0; rep < 3; ++rep)
#pragma unroll
          for (int rr = 0; rr < 3; ++rr)
#pragma unroll
            for (int c = 0; c < 6; ++c) {
              acc[c] = fma(rep == 2 ? f[rr * 6 + c] : m[rr * 6 + c], b[rr], acc[c]);
              acc[6 + c] = fma(rep == 2 ? f[(rr + 3) * 6 + c - 2] : m[(rr + 3) * 6 + c], b[rr + 3], acc[6 + c]);
            }
#pragma unroll
        for (int k = 0; k < 6; ++k) { const double v = acc[k] + acc[6 + k]; vec[t * 6 + k] = v; pr[t * 6 + k] = v * 0.5; a[k] = v * 1e-3 + 1.0; }
      }
    } else if (VARIANT == 5) {          // every solver lane busy: 36 multiply-adds each (multi-lane layout), LDS read-modify-write
      if (solver) {
        const int node = (t & ~1) | 1;
        double b[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = vec[node * 6 + k];
        double acc[12];
#pragma unroll
        for (int q = 0; q < 12; ++q) acc[q] = 0.0;
#pragma unroll
        for (int rr = 0; rr < 3; ++rr)
#pragma unroll
          for (int c = 0; c < 6; ++c) {
            acc[c] = fma(m[rr * 6 + c], b[rr], acc[c]);
            acc[6 + c] = fma(m[(rr + 3) * 6 + c], b[rr + 3], acc[6 + c]);
          }
#pragma unroll
        for (int k = 0; k < 6; ++k) pr[((t ^ 1) & 255) * 6 + k] += acc[k] + acc[6 + k];
      }
    }
    __syncthreads();
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 256 && blockIdx.x == 0) ticks[VARIANT] = t1 - t0;
  if (tid == 0 && blockIdx.x == 0) ticks[8 + VARIANT] = t1 - t0;
  double s = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) s += a[k];
  out[blockIdx.x * 512 + tid] = s + m[tid % 36];
}
int main(int argc, char** argv) {
  const int grid = argc > 1 ? atoi(argv[1]) : 1, rounds = 2000;
  double* out; long long* ticks;
  hipMalloc(&out, (size_t)grid * 512 * 8); hipMalloc(&ticks, 16 * 8);
  const char* names[6] = {"barrier only", "LDS exchange (12 read, 12 written)", "exchange + 108 fma (registers)", "exchange + 36 fma",
                          "exchange + 34 doubles factor from LDS + 108 fma", "all lanes: 36 fma + LDS read-modify-write"};
  for (int h = 1; h <= 16; h *= 4) {
    printf("grid %d, active solver lanes: 1 of %d\n", grid, 2 * h);
#define RUN(V) { hipFuncSetAttribute((const void*)step_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000); \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(step_kernel<V>, dim3(grid), dim3(512), 140000, 0, out, ticks, rounds, h); hipDeviceSynchronize(); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5)
    long long hh[16]; hipMemcpy(hh, ticks, 128, hipMemcpyDeviceToHost);
    for (int v = 0; v < 6; ++v) printf("  %-52s %7.0f cycles per step (solver wave), %7.0f (row wave)\n", names[v], hh[v] / (double)rounds, hh[8 + v] / (double)rounds);
  }
  return 0;
}
