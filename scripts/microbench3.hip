// Calibration for the register-exchange BCR levels (diagnostic, not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off microbench3.hip -o microbench3.bin
// Part 1: fp64 FMA dependent-chain latency / issue rate (K independent chains).
// Part 2: cross-lane moves of a 6-vector of doubles: DPP (v_mov_b32_dpp x 12), ds_bpermute_b32 x 12, LDS write + read.
// Part 3: one BCR level, old form (72 FMAs on the node's lane, partials through LDS, workgroup barrier) against the pair-split
//         form (36 FMAs on the node's lane and 36 on its neighbour, operands and partials through DPP / ds_bpermute, no barrier).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ double dpp_f64(double v, const int ctrl_sel) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  switch (ctrl_sel) {   // compile-time after inlining
    case 0: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xF5, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xF5, 0xF, 0xF, false); break;   // quad_perm [1,1,3,3]
    case 1: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xA0, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xA0, 0xF, 0xF, false); break;   // quad_perm [0,0,2,2]
    case 2: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xF, 0xF, false); break; // wave_shr:1
    case 3: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x102, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x102, 0xF, 0xF, false); break; // row_shl:2
    case 4: lo = __builtin_amdgcn_update_dpp(lo, lo, 0x130, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x130, 0xF, 0xF, false); break; // wave_shl:1
    case 5: lo = __builtin_amdgcn_update_dpp(lo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0xB1, 0xF, 0xF, false); break;   // quad_perm [1,0,3,2]
  }
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dppz_f64(double v) {   // invalid source lanes read 0 (bound_ctrl): no copy of the old value
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bperm_f64(double v, const int src_lane) {
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}

// ---- part 1 ----
template <int K>
__global__ __launch_bounds__(64) void fma_chain(double* out, long long* ticks, int n) {
  double a[K], m = 1.0 + 1e-9 * threadIdx.x, c = 1e-12;
#pragma unroll
  for (int k = 0; k < K; ++k) a[k] = 1.0 + k;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int rep = 0; rep < 8; ++rep)
#pragma unroll
      for (int k = 0; k < K; ++k) a[k] = fma(a[k], m, c);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s += a[k];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}

// ---- part 2: move a 6-vector from a neighbouring lane, dependent on the previous round's arithmetic ----
template <int HOW>
__global__ __launch_bounds__(64) void move6(double* out, long long* ticks, int n, double* scratch) {
  __shared__ double lds[64 * 6 + 16];
  double b[6];
  const int lane = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 6; ++k) b[k] = 1.0 + k + 1e-3 * lane;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    double v[6];
    if (HOW == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = dpp_f64(b[k], 2);
    } else if (HOW == 1) {
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = bperm_f64(b[k], (lane + 63) & 63);
    } else if (HOW == 2) {
#pragma unroll
      for (int k = 0; k < 6; ++k) lds[lane * 6 + k] = b[k];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = lds[((lane + 63) & 63) * 6 + k];
    } else {   // nothing moved: the arithmetic alone
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = b[k];
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) b[k] = b[k] * 0.5 + v[k] * 0.25;   // one dependent mul + one dependent add... (two rounded ops)
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) s += b[k];
  out[lane] = s;
  if (lane == 0) ticks[0] = t1 - t0;
}

// ---- part 3 ----
// 512 threads: 4 idle "row" waves that only meet the barriers + 4 solver waves.  LEVELS levels per round, one barrier per round
// in the new forms (the barrier in front of the tail), one barrier per level in the old form.
template <int VARIANT>
__global__ __launch_bounds__(512) void level_kernel(double* out, long long* ticks, int rounds, int levels) {
  extern __shared__ __align__(16) double lds[];
  const int tid = threadIdx.x;
  for (int k = tid; k < 16384; k += blockDim.x) lds[k] = 1.0 + 1e-9 * k;
  double A[36], Bm[36], b[6];
#pragma unroll
  for (int k = 0; k < 36; ++k) { A[k] = 1e-3 * (1.0 + 1e-3 * (k + tid)); Bm[k] = 1e-3 * (1.0 + 1e-3 * (2 * k + tid)); }
#pragma unroll
  for (int k = 0; k < 6; ++k) b[k] = 1.0 + k;
  __syncthreads();
  const bool solver = tid >= 256;
  const int t = tid - 256, lane = tid & 63;
  double* vec = lds;            // [256][6]
  double* pr = lds + 6 * 256;   // [256][6]
  double* fx = lds + 12 * 256;  // [256][12]
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < rounds; ++r) {
    if (VARIANT == 0) {   // old: per level 72 FMAs on the eliminated lane (12 of F_r's entries from LDS), partials through LDS, barrier, absorb
      for (int lv = 0, h = 1; lv < levels; ++lv, h <<= 1) {
        const int m2 = 2 * h - 1;
        if (solver) {
          if (h > 1 && (t & (h - 1)) == 0) {
            const int hp = h >> 1;
            if (t >= hp)
#pragma unroll
              for (int k = 0; k < 6; ++k) b[k] -= pr[(t - hp) * 6 + k];
            if (t + hp < 256)
#pragma unroll
              for (int k = 0; k < 6; ++k) b[k] -= vec[(t + hp) * 6 + k];
          }
          if ((t & m2) == h) {
            double pa[6] = {0, 0, 0, 0, 0, 0}, pb[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
              for (int c = 0; c < 6; ++c) {
                pa[c] = fma(A[rr * 6 + c], b[rr], pa[c]);
                pb[c] = fma(A[(rr + 3) * 6 + c], b[rr + 3], pb[c]);
              }
#pragma unroll
            for (int c = 0; c < 6; ++c) vec[t * 6 + c] = pa[c] + pb[c];
            double er[36];
#pragma unroll
            for (int k = 0; k < 24; ++k) er[k] = Bm[k];
#pragma unroll
            for (int k = 0; k < 12; ++k) er[24 + k] = fx[t * 12 + k];
            double qa[6] = {0, 0, 0, 0, 0, 0}, qb[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int a = 0; a < 6; ++a) {
                qa[a] = fma(er[a * 6 + k], b[k], qa[a]);
                qb[a] = fma(er[a * 6 + k + 3], b[k + 3], qb[a]);
              }
#pragma unroll
            for (int a = 0; a < 6; ++a) pr[t * 6 + a] = qa[a] + qb[a];
          }
        }
        __syncthreads();
      }
      if (solver)
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = b[k] * 1e-3 + 1.0;
    } else {
      // new: every level, the node's lane and its neighbour each form one 36-FMA product of the node's rhs; operands and
      // partials move through registers.  VARIANT 1: ds_bpermute for the partials; 2: DPP only (the h = 2 pattern at every level);
      // 3: like 1 with a third of the second matrix from the lane's own LDS column (register budget)
      if (solver) {
        for (int lv = 0, h = 1; lv < levels; ++lv, h <<= 1) {
          const int m2 = 2 * h - 1;
          double v[6];
          if (VARIANT == 4) {
            if (lv == 0) {
#pragma unroll
              for (int k = 0; k < 6; ++k) v[k] = dppz_f64<0xF5>(b[k]);
            } else {
#pragma unroll
              for (int k = 0; k < 6; ++k) v[k] = dppz_f64<0xA0>(b[k]);
            }
          } else if (lv == 0) {
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = dpp_f64(b[k], 0);
          } else {
#pragma unroll
            for (int k = 0; k < 6; ++k) v[k] = dpp_f64(b[k], 1);
          }
          double qa[6] = {0, 0, 0, 0, 0, 0}, qb[6] = {0, 0, 0, 0, 0, 0};
          if (lv == 0) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int a = 0; a < 6; ++a) {
                qa[a] = fma(A[a * 6 + k], v[k], qa[a]);
                qb[a] = fma(A[a * 6 + k + 3], v[k + 3], qb[a]);
              }
          } else {
            double m[36];
#pragma unroll
            for (int k = 0; k < 36; ++k) m[k] = Bm[k];
            if (VARIANT == 3) {
#pragma unroll
              for (int k = 0; k < 12; ++k) m[24 + k] = fx[t * 12 + k];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
              for (int a = 0; a < 6; ++a) {
                qa[a] = fma(m[a * 6 + k], v[k], qa[a]);
                qb[a] = fma(m[a * 6 + k + 3], v[k + 3], qb[a]);
              }
          }
          double o[6], left[6], right[6];
#pragma unroll
          for (int a = 0; a < 6; ++a) o[a] = qa[a] + qb[a];
          if (VARIANT == 4) {
#pragma unroll
            for (int a = 0; a < 6; ++a) left[a] = dppz_f64<0x138>(o[a]);
            if (lv == 0) {
#pragma unroll
              for (int a = 0; a < 6; ++a) right[a] = o[a];
            } else {
#pragma unroll
              for (int a = 0; a < 6; ++a) right[a] = dppz_f64<0x102>(o[a]);
            }
          } else if (VARIANT == 2 || lv == 0) {
#pragma unroll
            for (int a = 0; a < 6; ++a) left[a] = dpp_f64(o[a], 2);
            if (lv == 0) {
#pragma unroll
              for (int a = 0; a < 6; ++a) right[a] = o[a];
            } else {
#pragma unroll
              for (int a = 0; a < 6; ++a) right[a] = dpp_f64(o[a], 3);
            }
          } else {
            const int sl = (lane - h + 1) & 63, sr = (lane + h) & 63;
#pragma unroll
            for (int a = 0; a < 6; ++a) left[a] = bperm_f64(o[a], sl);
#pragma unroll
            for (int a = 0; a < 6; ++a) right[a] = bperm_f64(o[a], sr);
          }
          if ((t & m2) == 0 && (lane != 0)) {
#pragma unroll
            for (int k = 0; k < 6; ++k) b[k] -= left[k];
            if (t + h < 250)
#pragma unroll
              for (int k = 0; k < 6; ++k) b[k] -= right[k];
          }
          if (lane == 0 && (t & m2) == 0) {   // wave-start node: deferred
#pragma unroll
            for (int k = 0; k < 6; ++k) pr[(t / 64 * 6 + lv) * 12 + k] = right[k];
          }
          if (lane == 64 - h + (lv == 0 ? 0 : 1)) {
#pragma unroll
            for (int k = 0; k < 6; ++k) pr[(t / 64 * 6 + lv) * 12 + 6 + k] = o[k];
          }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) b[k] = b[k] * 1e-3 + 1.0;
      }
      __syncthreads();
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 256 && blockIdx.x == 0) ticks[0] = t1 - t0;
  double s = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k) s += b[k];
#pragma unroll
  for (int k = 0; k < 36; ++k) s += A[k] + Bm[k];
  out[blockIdx.x * 512 + tid] = s;
}

// correctness probe of the DPP controls on this hardware
__global__ void dpp_probe(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x138, 0xF, 0xF, false);        // wave_shr:1
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x130, 0xF, 0xF, false);   // wave_shl:1
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x102, 0xF, 0xF, false);  // row_shl:2
  out[192 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x112, 0xF, 0xF, false);  // row_shr:2
  out[256 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0xF5, 0xF, 0xF, false);   // quad_perm [1,1,3,3]
  out[320 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0xA0, 0xF, 0xF, false);   // quad_perm [0,0,2,2]
  out[384 + lane] = __builtin_amdgcn_update_dpp(-1, lane, 0x142, 0xF, 0xF, false);  // row_bcast:15
  out[448 + lane] = __builtin_amdgcn_ds_bpermute(((lane + 17) & 63) << 2, lane);
}

int main(int argc, char** argv) {
  double* out; long long* ticks; double* scratch; int* probe;
  hipMalloc(&out, 1 << 20); hipMalloc(&ticks, 128); hipMalloc(&scratch, 1 << 16); hipMalloc(&probe, 512 * 4);
  long long hh[16];
  {
    hipLaunchKernelGGL(dpp_probe, dim3(1), dim3(64), 0, 0, probe);
    int hp[512]; hipMemcpy(hp, probe, sizeof(hp), hipMemcpyDeviceToHost);
    const char* nm[8] = {"wave_shr:1", "wave_shl:1", "row_shl:2", "row_shr:2", "quad_perm[1,1,3,3]", "quad_perm[0,0,2,2]", "row_bcast:15", "bpermute(+17)"};
    for (int q = 0; q < 8; ++q) {
      printf("%-20s", nm[q]);
      for (int l = 0; l < 64; ++l) if (l < 20 || l > 60) printf(" %d", hp[64 * q + l]);
      printf("\n");
    }
  }
  const int n = 4000;
#define CHAIN(K) { for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(fma_chain<K>, dim3(1), dim3(64), 0, 0, out, ticks, n); hipDeviceSynchronize(); \
    hipMemcpy(hh, ticks, 8, hipMemcpyDeviceToHost); printf("fma chains K=%2d: %.2f cycles per FMA, %.1f per round of K\n", K, hh[0] / (double)(n * 8 * K), hh[0] / (double)(n * 8)); }
  CHAIN(1) CHAIN(2) CHAIN(4) CHAIN(6) CHAIN(8) CHAIN(12) CHAIN(16)
#define MOVE(H, name) { for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(move6<H>, dim3(1), dim3(64), 0, 0, out, ticks, n, scratch); hipDeviceSynchronize(); \
    hipMemcpy(hh, ticks, 8, hipMemcpyDeviceToHost); printf("move a 6-vector by %-28s %.1f cycles per round\n", name, hh[0] / (double)n); }
  MOVE(3, "nothing (arithmetic only):") MOVE(0, "DPP wave_shr:1:") MOVE(1, "ds_bpermute:") MOVE(2, "LDS write + read:")
  const int rounds = 2000;
  const int grid = argc > 1 ? atoi(argv[1]) : 1;
  const char* names[5] = {"old: 72 FMA on one lane, LDS + barrier per level", "pair split, ds_bpermute partials", "pair split, DPP only", "pair split, ds_bpermute, 12 of 36 from LDS", "pair split, DPP only, bound_ctrl (no copies)"};
  for (int levels = 1; levels <= 5; levels += 4) {
#define RUN(V) { hipFuncSetAttribute((const void*)level_kernel<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 140000); \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float ms = 0; \
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0, 0); hipLaunchKernelGGL(level_kernel<V>, dim3(grid), dim3(512), 140000, 0, out, ticks, rounds, levels); hipEventRecord(e1, 0); hipDeviceSynchronize(); hipEventElapsedTime(&ms, e0, e1); } \
    hipMemcpy(hh, ticks, 8, hipMemcpyDeviceToHost); printf("  levels %d  %-52s %7.0f cycles per round, %6.0f per level   (kernel %.3f ms: %.2f ticks per ns)\n", levels, names[V], hh[0] / (double)rounds, hh[0] / (double)rounds / levels, ms, hh[0] / (ms * 1e6)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4)
  }
  return 0;
}
