// Calibration: cost of a BCR-level-like LDS read burst (57 doubles per lane as ds_read_b128) as a function of the
// number of waves issuing it and of the fraction of active lanes (diagnostic, not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MASK, int WAVES, bool FMA>
__device__ long long burst(const double* lds, int tid, double* sink) {
  const int t = tid - 256;
  const double* ps = lds + (t < 0 ? 0 : t) * 22;
  const double* pe = lds + 22 * 200 + (t < 0 ? 0 : t) * 38;
  double acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int round = 0; round < 100; ++round) {
    if (t >= 0 && t < 64 * WAVES && (t & MASK) == ((MASK + 1) >> 1)) {
      double v[58];
      #pragma unroll
      for (int k = 0; k < 22; ++k) v[k] = ps[k];
      #pragma unroll
      for (int k = 0; k < 36; ++k) v[22 + k] = pe[k];
      if (FMA) {
        #pragma unroll
        for (int k = 0; k < 58; ++k) acc[k % 12] = fma(v[k], acc[(k + 1) % 12] + 1.0, acc[k % 12]);
        #pragma unroll
        for (int k = 0; k < 50; ++k) acc[k % 12] = fma(v[k], acc[(k + 5) % 12], acc[k % 12]);
      } else {
        #pragma unroll
        for (int k = 0; k < 58; ++k) asm volatile("" ::"v"(v[k]));
      }
    }
    __syncthreads();
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
  for (int k = 0; k < 12; ++k) s += acc[k];
  if (s == 12345.678) sink[tid] = s;
  return t1 - t0;
}
__global__ __launch_bounds__(512) void mb(double* out, long long* ticks) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  for (int k = tid; k < 200 * 60; k += blockDim.x) lds[k] = 1e-3 * (k % 17);
  long long r[12];
  r[0] = burst<0, 3, false>(lds, tid, out);
  r[1] = burst<1, 3, false>(lds, tid, out);
  r[2] = burst<7, 3, false>(lds, tid, out);
  r[3] = burst<31, 3, false>(lds, tid, out);
  r[4] = burst<0, 1, false>(lds, tid, out);
  r[5] = burst<7, 1, false>(lds, tid, out);
  r[6] = burst<0, 3, true>(lds, tid, out);
  r[7] = burst<1, 3, true>(lds, tid, out);
  r[8] = burst<7, 3, true>(lds, tid, out);
  r[9] = burst<0, 1, true>(lds, tid, out);
  r[10] = burst<7, 1, true>(lds, tid, out);
  r[11] = burst<31, 1, true>(lds, tid, out);
  if (tid == 0) for (int k = 0; k < 12; ++k) ticks[k] = r[k];
}
int main() {
  double* out; long long* ticks;
  hipMalloc(&out, 512 * 8); hipMalloc(&ticks, 16 * 8);
  hipFuncSetAttribute((const void*)mb, hipFuncAttributeMaxDynamicSharedMemorySize, 120000);
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(mb, dim3(1), dim3(512), 100000, 0, out, ticks);
  hipDeviceSynchronize();
  long long h[16]; hipMemcpy(h, ticks, 12 * 8, hipMemcpyDeviceToHost);
  const char* names[12] = {"3 waves all lanes, loads only", "3 waves 1/2 lanes, loads only", "3 waves 1/8 lanes, loads only",
    "3 waves 1/32 lanes, loads only", "1 wave all lanes, loads only", "1 wave 1/8 lanes, loads only",
    "3 waves all lanes, loads+108 FMA", "3 waves 1/2 lanes, loads+FMA", "3 waves 1/8 lanes, loads+FMA",
    "1 wave all lanes, loads+FMA", "1 wave 1/8 lanes, loads+FMA", "1 wave 1/32 lanes, loads+FMA"};
  for (int k = 0; k < 12; ++k) printf("%-36s %6.0f cycles per round\n", names[k], h[k] * 24.0 / 100.0);
  return 0;
}
