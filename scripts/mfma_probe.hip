// Layout probe of v_mfma_f64_16x16x4_f64 on gfx950 (diagnostic): where A[i][k], B[k][j] and D[i][j] live.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef double double4_t __attribute__((ext_vector_type(4)));
__global__ void probe(double* out) {
  const int l = threadIdx.x;
  // hypothesis for the inputs: A[i][k] in lane i + 16 k, B[k][j] in lane j + 16 k
  const int i = l & 15, k = l >> 4;
  const double a = 1.0 + i + 0.01 * k;          // A[i][k]
  const double b = 100.0 * (i + 1) + 7.0 * k;   // B[k][j] with j = l & 15
  double4_t c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
int main() {
  double* d; hipMalloc(&d, 64 * 4 * 8);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  double h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  // reference D[i][j] = sum_k A[i][k] B[k][j]
  double D[16][16];
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += (1.0 + i + 0.01 * k) * (100.0 * (j + 1) + 7.0 * k); D[i][j] = s; }
  int ok = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) if (std::fabs(D[i][j] - h[l * 4 + r]) < 1e-9 * std::fabs(D[i][j])) {
      if (l < 20 || l > 60) printf("lane %2d reg %d = D[%2d][%2d]\n", l, r, i, j);
      ok++;
    }
  }
  printf("matched %d of 256\n", ok);
  return 0;
}
