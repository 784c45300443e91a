// MFMA experiment of round 4 (diagnostic, not part of the library): the one dense piece of the DO backend - the in-place inversion
// of the <= 36 x 36 SPD tail system of a BCR factorisation (csrc/dsqp_program_impl.h: bcr_factor, "dense tail") -
//   A  as the kernel does it: block Gauss-Jordan with 6 x 6 pivot blocks on the matrix in LDS, 256 threads, 4 barriers per pivot;
//   B  with the matrix cores: ONE wave holds the matrix (padded to 48 x 48) in registers as 3 x 3 accumulator tiles of
//      v_mfma_f64_16x16x4_f64, block Gauss-Jordan with 4 x 4 pivot blocks: per pivot the rank-4 update of all nine tiles is nine
//      MFMAs (A operand: the pivot columns, B operand: the scaled pivot rows), pivot rows / columns travel through LDS.
// Both are checked against a host inverse; cycles by s_memtime.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off microbench_mfma_tail.hip -o microbench_mfma_tail.bin
// Register layout of v_mfma_f64_16x16x4_f64 on gfx950 (scripts/mfma_probe.hip): A[i][k] in lane i + 16 k, B[k][j] in lane
// j + 16 k, D[i][j] in lane j + 16 (i % 4), register i / 4.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
constexpr int N = 36, LD = 38, NP = 48;

// ---------------------------------------------------------------- variant A: LDS block Gauss-Jordan, 6 x 6 pivots
__device__ __forceinline__ constexpr int sym(int r, int c) { return r >= c ? r * (r + 1) / 2 + c : c * (c + 1) / 2 + r; }
__device__ void spd_inverse6(const double (&A)[21], double (&inv)[21]) {
  double L[6][6], d[6], dinv[6], M[6][6];
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    double dj = A[sym(j, j)];
#pragma unroll
    for (int k = 0; k < j; ++k) dj = fma(-L[j][k] * d[k], L[j][k], dj);
    d[j] = dj;
    dinv[j] = 1.0 / dj;
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = A[sym(i, j)];
#pragma unroll
      for (int k = 0; k < j; ++k) v = fma(-L[i][k] * d[k], L[j][k], v);
      L[i][j] = v * dinv[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 6; ++j)
#pragma unroll
    for (int i = j + 1; i < 6; ++i) {
      double v = -L[i][j];
#pragma unroll
      for (int k = j + 1; k < i; ++k) v = fma(-L[i][k], M[k][j], v);
      M[i][j] = v;
    }
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double v = (r == c) ? dinv[r] : M[r][c] * dinv[r];
#pragma unroll
      for (int k = r + 1; k < 6; ++k) v = fma(M[k][r] * dinv[k], M[k][c], v);
      inv[sym(r, c)] = v;
    }
}
__global__ __launch_bounds__(512) void invert_lds(const double* __restrict__ A, double* __restrict__ out, long long* ticks, int rounds) {
  __shared__ double t[N * LD];
  __shared__ double pinv[36];
  const int tid = threadIdx.x - 256;   // the solver half of a 512-thread workgroup does it, the other half meets the barriers
  const int nthr = 256;
  long long acc = 0;
  for (int rd = 0; rd < rounds; ++rd) {
    if (tid >= 0)
      for (int e = tid; e < N * N; e += nthr) t[(e / N) * LD + e % N] = A[e];
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    auto pivot_inverse = [&](const int p) {
      double Ain[21], Pin[21];
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c <= r; ++c) Ain[sym(r, c)] = t[(6 * p + r) * LD + 6 * p + c];
      spd_inverse6(Ain, Pin);
#pragma unroll
      for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) pinv[r * 6 + c] = Pin[sym(r, c)];
    };
    if (tid == 0) pivot_inverse(0);
    __syncthreads();
    for (int p = 0; p < 6; ++p) {
      const int p0 = 6 * p;
      if (tid >= 0)
        for (int c = tid; c < N; c += nthr)
          if (c < p0 || c >= p0 + 6) {
            double a[6], nw[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = t[(p0 + j) * LD + c];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              double v = 0.0;
#pragma unroll
              for (int j = 0; j < 6; ++j) v = fma(pinv[k * 6 + j], a[j], v);
              nw[k] = v;
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) t[(p0 + k) * LD + c] = nw[k];
          }
      __syncthreads();
      if (tid >= 0)
        for (int e = tid; e < N * N; e += nthr) {
          const int r = e / N, c = e - r * N;
          if ((r < p0 || r >= p0 + 6) && (c < p0 || c >= p0 + 6)) {
            double v = t[r * LD + c];
#pragma unroll
            for (int k = 0; k < 6; ++k) v = fma(-t[r * LD + p0 + k], t[(p0 + k) * LD + c], v);
            t[r * LD + c] = v;
          }
        }
      __syncthreads();
      if (tid >= 0)
        for (int r = tid; r < N; r += nthr) {
          double nw[6];
          if (r < p0 || r >= p0 + 6) {
            double a[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) a[j] = t[r * LD + p0 + j];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
              double v = 0.0;
#pragma unroll
              for (int j = 0; j < 6; ++j) v = fma(-a[j], pinv[j * 6 + k], v);
              nw[k] = v;
            }
          } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) nw[k] = pinv[(r - p0) * 6 + k];
          }
#pragma unroll
          for (int k = 0; k < 6; ++k) t[r * LD + p0 + k] = nw[k];
        }
      __syncthreads();
      if (p + 1 < 6) {
        if (tid == 0) pivot_inverse(p + 1);
        __syncthreads();
      }
    }
    acc += __builtin_amdgcn_s_memtime() - t0;
  }
  if (tid >= 0)
    for (int e = tid; e < N * N; e += nthr) out[e] = t[(e / N) * LD + e % N];
  if (tid == 0) ticks[0] = acc;
}

// ---------------------------------------------------------------- variant B: one wave, matrix in MFMA accumulator tiles
__global__ __launch_bounds__(64) void invert_mfma(const double* __restrict__ A, double* __restrict__ out, long long* ticks, int rounds) {
  __shared__ double rowb[4 * NP];    // pivot rows  [q][J]
  __shared__ double colb[NP * 4];    // pivot cols  [I][s]
  __shared__ double row2[4 * NP];    // scaled pivot rows (Pinv in the pivot block)
  __shared__ double col2[NP * 4];    // new pivot cols
  __shared__ double pinv[16];
  const int l = threadIdx.x, lj = l & 15, lq = l >> 4;
  long long acc = 0;
  double4_t T[3][3];
  for (int rd = 0; rd < rounds; ++rd) {
#pragma unroll
    for (int bi = 0; bi < 3; ++bi)
#pragma unroll
      for (int bj = 0; bj < 3; ++bj)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int I = 16 * bi + 4 * r + lq, J = 16 * bj + lj;
          T[bi][bj][r] = (I < N && J < N) ? A[I * N + J] : (I == J ? 1.0 : 0.0);
        }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < 9; ++p) {          // (the padding block is the identity: nothing to do for pivots 9 .. 11)
      const int pb = p / 4, pr = p % 4, P0 = 4 * p;
      // pivot rows: register pr of the tiles of tile row pb, all lanes; pivot columns: the lanes whose column is in the block
#pragma unroll
      for (int bj = 0; bj < 3; ++bj) rowb[lq * NP + 16 * bj + lj] = T[pb][bj][pr];
      if ((lj >> 2) == pr) {
#pragma unroll
        for (int bi = 0; bi < 3; ++bi)
#pragma unroll
          for (int r = 0; r < 4; ++r) colb[(16 * bi + 4 * r + lq) * 4 + (lj & 3)] = T[bi][pb][r];
      }
      __syncthreads();
      if (l == 0) {    // 4 x 4 SPD pivot block: LDL' inverse on one lane
        double a[4][4], L[4][4], d[4], di[4], M[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < 4; ++c) a[r][c] = rowb[r * NP + P0 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          double dj = a[j][j];
#pragma unroll
          for (int k = 0; k < j; ++k) dj = fma(-L[j][k] * d[k], L[j][k], dj);
          d[j] = dj;
          di[j] = 1.0 / dj;
#pragma unroll
          for (int i = j + 1; i < 4; ++i) {
            double v = a[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) v = fma(-L[i][k] * d[k], L[j][k], v);
            L[i][j] = v * di[j];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = j + 1; i < 4; ++i) {
            double v = -L[i][j];
#pragma unroll
            for (int k = j + 1; k < i; ++k) v = fma(-L[i][k], M[k][j], v);
            M[i][j] = v;
          }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c <= r; ++c) {
            double v = (r == c) ? di[r] : M[r][c] * di[r];
#pragma unroll
            for (int k = r + 1; k < 4; ++k) v = fma(M[k][r] * di[k], M[k][c], v);
            pinv[r * 4 + c] = v;
            pinv[c * 4 + r] = v;
          }
      }
      __syncthreads();
      // scaled pivot rows (Pinv itself inside the pivot block) and new pivot columns, three entries per lane each
#pragma unroll
      for (int bj = 0; bj < 3; ++bj) {
        const int J = 16 * bj + lj;
        double v = 0.0;
#pragma unroll
        for (int s = 0; s < 4; ++s) v = fma(pinv[lq * 4 + s], rowb[s * NP + J], v);
        row2[lq * NP + J] = (J >= P0 && J < P0 + 4) ? pinv[lq * 4 + (J - P0)] : v;
      }
#pragma unroll
      for (int k3 = 0; k3 < 3; ++k3) {
        const int I = 16 * k3 + lj;       // (lane (lj, lq): row I, column lq of the block)
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) v = fma(-colb[I * 4 + q], pinv[q * 4 + lq], v);
        col2[I * 4 + lq] = v;
      }
      __syncthreads();
      // rank-4 update of all nine tiles: T -= C R' (rows / columns of the pivot block are overwritten below)
      double a_op[3], b_op[3];
#pragma unroll
      for (int bi = 0; bi < 3; ++bi) a_op[bi] = -colb[(16 * bi + lj) * 4 + lq];
#pragma unroll
      for (int bj = 0; bj < 3; ++bj) b_op[bj] = row2[lq * NP + 16 * bj + lj];
#pragma unroll
      for (int bi = 0; bi < 3; ++bi)
#pragma unroll
        for (int bj = 0; bj < 3; ++bj) T[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_op[bi], b_op[bj], T[bi][bj], 0, 0, 0);
      if ((lj >> 2) == pr) {
#pragma unroll
        for (int bi = 0; bi < 3; ++bi)
#pragma unroll
          for (int r = 0; r < 4; ++r) T[bi][pb][r] = col2[(16 * bi + 4 * r + lq) * 4 + (lj & 3)];
      }
#pragma unroll
      for (int bj = 0; bj < 3; ++bj) T[pb][bj][pr] = row2[lq * NP + 16 * bj + lj];
      __syncthreads();
    }
    acc += __builtin_amdgcn_s_memtime() - t0;
  }
#pragma unroll
  for (int bi = 0; bi < 3; ++bi)
#pragma unroll
    for (int bj = 0; bj < 3; ++bj)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int I = 16 * bi + 4 * r + lq, J = 16 * bj + lj;
        if (I < N && J < N) out[I * N + J] = T[bi][bj][r];
      }
  if (l == 0) ticks[0] = acc;
}

int main() {
  // SPD block-tridiagonal test matrix with 6 x 6 blocks (the tail's structure), diagonally dominant
  std::vector<double> A(N * N, 0.0), ref(N * N);
  srand(7);
  for (int r = 0; r < N; ++r)
    for (int c = 0; c <= r; ++c)
      if (r / 6 - c / 6 <= 1) {
        const double v = (r == c) ? 8.0 + (rand() % 100) * 0.01 : ((rand() % 200) - 100) * 0.004;
        A[r * N + c] = A[c * N + r] = v;
      }
  {  // host inverse (Gauss-Jordan with partial pivoting, long double)
    std::vector<long double> M(N * 2 * N, 0.0L);
    for (int r = 0; r < N; ++r) {
      for (int c = 0; c < N; ++c) M[r * 2 * N + c] = A[r * N + c];
      M[r * 2 * N + N + r] = 1.0L;
    }
    for (int p = 0; p < N; ++p) {
      int best = p;
      for (int r = p + 1; r < N; ++r)
        if (fabsl(M[r * 2 * N + p]) > fabsl(M[best * 2 * N + p])) best = r;
      for (int c = 0; c < 2 * N; ++c) std::swap(M[p * 2 * N + c], M[best * 2 * N + c]);
      const long double d = M[p * 2 * N + p];
      for (int c = 0; c < 2 * N; ++c) M[p * 2 * N + c] /= d;
      for (int r = 0; r < N; ++r)
        if (r != p) {
          const long double f = M[r * 2 * N + p];
          for (int c = 0; c < 2 * N; ++c) M[r * 2 * N + c] -= f * M[p * 2 * N + c];
        }
    }
    for (int r = 0; r < N; ++r)
      for (int c = 0; c < N; ++c) ref[r * N + c] = (double)M[r * 2 * N + N + c];
  }
  double *dA, *dOut;
  long long* dT;
  hipMalloc(&dA, N * N * 8);
  hipMalloc(&dOut, N * N * 8);
  hipMalloc(&dT, 64);
  hipMemcpy(dA, A.data(), N * N * 8, hipMemcpyHostToDevice);
  const int rounds = 200;
  std::vector<double> got(N * N);
  long long tk;
  for (int v = 0; v < 2; ++v) {
    for (int rep = 0; rep < 2; ++rep) {
      if (v == 0) hipLaunchKernelGGL(invert_lds, dim3(1), dim3(512), 0, 0, dA, dOut, dT, rounds);
      else hipLaunchKernelGGL(invert_mfma, dim3(1), dim3(64), 0, 0, dA, dOut, dT, rounds);
      hipDeviceSynchronize();
    }
    hipMemcpy(got.data(), dOut, N * N * 8, hipMemcpyDeviceToHost);
    hipMemcpy(&tk, dT, 8, hipMemcpyDeviceToHost);
    double err = 0, mx = 0;
    for (int e = 0; e < N * N; ++e) {
      err = std::fmax(err, std::fabs(got[e] - ref[e]));
      mx = std::fmax(mx, std::fabs(ref[e]));
    }
    printf("%-58s %8.0f cycles per inversion   max |error| %.2e (max |entry| %.2e)\n",
           v == 0 ? "A  LDS block Gauss-Jordan, 6x6 pivots, 256 threads:" : "B  v_mfma_f64_16x16x4_f64, 4x4 pivots, one wave (81 MFMAs):",
           tk / (double)rounds, err, mx);
  }
  return 0;
}
