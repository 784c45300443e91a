// Instruction issue costs of one wave alone on its SIMD (diagnostic): cycles per instruction for unrolled streams.
//   hipcc --offload-arch=gfx950 -O3 microbench4.hip -o microbench4.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int WHAT>
__global__ __launch_bounds__(64) void k(long long* ticks, double* out, int n) {
  double a0 = 1.0 + threadIdx.x, a1 = 2.0, a2 = 3.0, a3 = 4.0, a4 = 5.0, a5 = 6.0, m = 1.0000001;
  int i0 = threadIdx.x, i1 = 2, i2 = 3, i3 = 4;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    if (WHAT == 0) {   // 64 independent-ish fp64 FMAs over 4 chains
      asm volatile(REP16("v_fma_f64 %0, %0, %4, %5\n v_fma_f64 %1, %1, %4, %5\n v_fma_f64 %2, %2, %4, %5\n v_fma_f64 %3, %3, %4, %5\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m), "v"(a4));
    } else if (WHAT == 1) {   // 64 v_mov_b32_dpp (4 registers round robin)
      asm volatile(REP16("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                         "v_mov_b32_dpp %2, %3 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %3, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    } else if (WHAT == 2) {   // 64 v_mov_b32_dpp quad_perm
      asm volatile(REP16("v_mov_b32_dpp %0, %1 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %1, %2 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:0\n"
                         "v_mov_b32_dpp %2, %3 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_mov_b32_dpp %3, %0 quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf bound_ctrl:0\n")
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    } else if (WHAT == 3) {   // 64 plain v_mov_b32
      asm volatile(REP16("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3));
    } else if (WHAT == 4) {   // 64 v_add_f64 over 4 chains
      asm volatile(REP16("v_add_f64 %0, %0, %4\n v_add_f64 %1, %1, %4\n v_add_f64 %2, %2, %4\n v_add_f64 %3, %3, %4\n")
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(m));
    } else if (WHAT == 5) {   // 16 exec-mask regions (saveexec + branch-not-taken + restore)
      asm volatile(REP16("v_cmp_eq_u32 vcc, 3, %0\n s_and_saveexec_b64 s[10:11], vcc\n s_cbranch_execz 1f\n v_add_f64 %1, %1, %2\n 1:\n s_or_b64 exec, exec, s[10:11]\n")
                   : "+v"(i0), "+v"(a0) : "v"(m) : "s10", "s11", "vcc");
    } else if (WHAT == 6) {   // 64 ds_bpermute_b32 in batches of 12 + wait
      asm volatile(REP4("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                        "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                        "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n"
                        "ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)\n")
                   : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"((int)((threadIdx.x + 1) & 63) << 2));
    } else if (WHAT == 7) {   // dependent: fma -> dpp (lo, hi) -> fma ... (latency of a DPP hop between dependent fp64 ops), 16 hops
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        a0 = fma(a0, m, m);
        const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a0), 0x138, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a0), 0x138, 0xF, 0xF, true);
        a0 = __hiloint2double(hi, lo);
      }
    } else if (WHAT == 8) {   // dependent fma only, 16
      asm volatile(REP16("v_fma_f64 %0, %0, %1, %1\n") : "+v"(a0) : "v"(m));
    } else if (WHAT == 9) {   // v_cndmask_b32 x 64
      asm volatile(REP16("v_cndmask_b32 %0, %1, %2, vcc\n v_cndmask_b32 %1, %2, %3, vcc\n v_cndmask_b32 %2, %3, %0, vcc\n v_cndmask_b32 %3, %0, %1, vcc\n") : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "vcc");
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3;
  if (threadIdx.x == 0) ticks[0] = t1 - t0;
}
int main() {
  long long* ticks; double* out; hipMalloc(&ticks, 64); hipMalloc(&out, 4096);
  const int n = 2000; long long h;
  const char* nm[10] = {"v_fma_f64 x64 (4 chains)", "v_mov_b32_dpp wave_shr x64", "v_mov_b32_dpp quad_perm x64", "v_mov_b32 x64", "v_add_f64 x64 (4 chains)",
                        "exec region x16 (cmp, saveexec, branch, add, restore)", "ds_bpermute_b32 x64 (16 per wait)", "fma -> dpp pair (dependent) x16", "fma dependent x16", "v_cndmask_b32 x64"};
  const int cnt[10] = {64, 64, 64, 64, 64, 16, 64, 16, 16, 64};
#define R(W) { for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(k<W>, dim3(1), dim3(64), 0, 0, ticks, out, n); hipDeviceSynchronize(); hipMemcpy(&h, ticks, 8, hipMemcpyDeviceToHost); \
  printf("%-56s %7.1f cycles per loop, %6.2f per item\n", nm[W], h / (double)n, h / (double)n / cnt[W]); }
  R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7) R(8) R(9)
  return 0;
}
