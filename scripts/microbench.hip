// Calibration micro-benchmarks for the BCR level design (diagnostic, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(512) void mb(double* out, long long* ticks, int nthreads_active) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  for (int k = tid; k < 8192; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
  __syncthreads();
  double a[6] = {1, 2, 3, 4, 5, 6}, m = 1.0000001, c = 1e-9;
  long long t0, t1;
  // (1) 216 FMAs in 6 independent chains
  t0 = __builtin_amdgcn_s_memtime();
  #pragma unroll
  for (int r = 0; r < 36; ++r) {
    #pragma unroll
    for (int k = 0; k < 6; ++k) a[k] = fma(a[k], m, c);
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[0] = t1 - t0;
  // (2) 36 FMAs one dependent chain
  t0 = __builtin_amdgcn_s_memtime();
  #pragma unroll
  for (int r = 0; r < 36; ++r) a[0] = fma(a[0], m, c);
  t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[1] = t1 - t0;
  // (3) 9 x ds_read_b128 (lane stride 38 doubles) + consume
  const double* p = lds + (tid % 200) * 38;
  double s = 0;
  t0 = __builtin_amdgcn_s_memtime();
  #pragma unroll
  for (int k = 0; k < 18; ++k) s += p[k];
  asm volatile("" :: "v"(s));
  t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[2] = t1 - t0;
  // (4) 100 barriers
  t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < 100; ++k) __syncthreads();
  t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[3] = t1 - t0;
  // (5) barrier + LDS write/read ping: 100 rounds of write -> sync -> read neighbour
  t0 = __builtin_amdgcn_s_memtime();
  double v = a[1];
  for (int k = 0; k < 100; ++k) {
    lds[tid] = v;
    __syncthreads();
    v = lds[(tid + 1) & 511] + 1.0;
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[4] = t1 - t0;
  // (6) only a subset of lanes does 108 FMAs between barriers (like a BCR level), 100 rounds
  t0 = __builtin_amdgcn_s_memtime();
  for (int k = 0; k < 100; ++k) {
    if (tid >= 256 && ((tid - 256) & 3) == 2) {
      #pragma unroll
      for (int r = 0; r < 18; ++r) {
        #pragma unroll
        for (int q = 0; q < 6; ++q) a[q] = fma(a[q], m, c);
      }
    }
    __syncthreads();
  }
  t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) ticks[5] = t1 - t0;
  out[tid] = a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + s + v;
}
int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 512 * 8); hipMalloc(&ticks, 8 * 8);
  hipFuncSetAttribute((const void*)mb, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mb, dim3(1), dim3(512), 65536, 0, out, ticks, 512);
  hipDeviceSynchronize();
  long long h[8]; hipMemcpy(h, ticks, 64, hipMemcpyDeviceToHost);
  printf("216 fma (6 chains): %lld cyc -> %.2f cyc/fma\n", h[0], h[0] / 216.0);
  printf("36 fma (1 chain):   %lld cyc -> %.2f cyc/fma\n", h[1], h[1] / 36.0);
  printf("18 LDS doubles read+sum: %lld cyc\n", h[2]);
  printf("barrier (512 thr): %.1f cyc each\n", h[3] / 100.0);
  printf("write+barrier+read neighbour: %.1f cyc per round\n", h[4] / 100.0);
  printf("level-like (108 fma on 1/4 of solver lanes + barrier): %.1f cyc per round\n", h[5] / 100.0);
  return 0;
}
