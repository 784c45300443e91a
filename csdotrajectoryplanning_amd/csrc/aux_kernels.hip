// aux_kernels.hip — the O(Nt Na^2) kernels either side of the solve:
//   K0  neighbour search + separating planes on the device: findNeighborPairsByTrustRegion and calcEqualInterPlanes
//       (sqp/inter_agent_cons.cc:12-49, 54-140 of the reference) on the float disc centres the host interpolation
//       produced.  Output order is the reference's (t, i, j); every operation is an IEEE add / multiply / sqrt / compare on
//       the same operands in the same order as bridge_host.cc, so pairs, legality flag and plane coefficients are
//       bit-identical to the host bridge (tests/test_aux_kernels.py).
//   validator  rectangle / rectangle and disc / rectangle checks of final trajectories, one lane per (t, i, j) and
//       (t, i, obstacle) (scripts/collision_detection.py:20-96 of the reference, pinned by tests/golden/ref_collision_verdicts.npz).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "aux_kernels.h"

namespace csdo {

namespace {

struct CentresDev {
  const float *xf, *yf, *xr, *yr, *xc, *yc, *cs, *sn;   // [Na][Nt]
};

__device__ __forceinline__ double sq_f(float a, float b) {   // pow(float - float, 2): float difference, double square
  const double d = (double)(a - b);
  return d * d;
}

// (t, i, j) is a neighbour pair: closest of the four disc-centre distances below 2 sqrt(2) r_trust (:31-37)
__device__ __forceinline__ bool near_pair(const CentresDev& C, size_t ki, size_t kj, double reach) {
  double d2 = sq_f(C.xf[ki], C.xf[kj]) + sq_f(C.yf[ki], C.yf[kj]);
  d2 = fmin(d2, sq_f(C.xf[ki], C.xr[kj]) + sq_f(C.yf[ki], C.yr[kj]));
  d2 = fmin(d2, sq_f(C.xr[ki], C.xf[kj]) + sq_f(C.yr[ki], C.yf[kj]));
  d2 = fmin(d2, sq_f(C.xr[ki], C.xr[kj]) + sq_f(C.yr[ki], C.yr[kj]));
  return sqrt(d2) < reach;
}

// State::agentCollision (common/motion_planning.h:140-183): rectangle SAT in float
__device__ __forceinline__ bool rect_hit(const CentresDev& C, size_t ki, size_t kj, float length, float width) {
  const float sx = C.xc[kj] - C.xc[ki], sy = C.yc[kj] - C.yc[ki];
  const float cv = C.cs[ki], sv = C.sn[ki], co = C.cs[kj], so = C.sn[kj];
  const float hl = length / 2, hw = width / 2;
  const float dx1 = cv * length / 2, dy1 = sv * length / 2, dx2 = sv * width / 2, dy2 = -cv * width / 2;
  const float dx3 = co * length / 2, dy3 = so * length / 2, dx4 = so * width / 2, dy4 = -co * width / 2;
  return (fabsf(sx * cv + sy * sv) <= fabsf(dx3 * cv + dy3 * sv) + fabsf(dx4 * cv + dy4 * sv) + hl) &&
         (fabsf(sx * sv - sy * cv) <= fabsf(dx3 * sv - dy3 * cv) + fabsf(dx4 * sv - dy4 * cv) + hw) &&
         (fabsf(sx * co + sy * so) <= fabsf(dx1 * co + dy1 * so) + fabsf(dx2 * co + dy2 * so) + hl) &&
         (fabsf(sx * so - sy * co) <= fabsf(dx1 * so - dy1 * co) + fabsf(dx2 * so - dy2 * co) + hw);
}

// One 64-lane workgroup per (t, i): lanes stride over j > i.  Pass 1 counts the pairs of (t, i) and flags rectangle
// overlaps; pass 2 (EMIT) writes them at the (t, i) block's offset in j order with the pair's eight planes.
template <bool EMIT>
__global__ __launch_bounds__(64) void k0_pairs_kernel(CentresDev C, int Na, int Nt, double reach, float length, float width,
                                                       double rv, int* __restrict__ counts, int* __restrict__ collide,
                                                       const long long* __restrict__ offsets, int32_t* __restrict__ pairs,
                                                       double* __restrict__ coef) {
  const int b = (int)blockIdx.x;            // b = t * Na + i: blocks in the reference's (t, i) order
  const int t = b / Na, i = b - t * Na;
  const int lane = (int)threadIdx.x;
  const size_t ki = (size_t)i * Nt + t;
  int total = 0;
  bool any_hit = false;
  for (int j0 = i + 1; j0 < Na; j0 += 64) {
    const int j = j0 + lane;
    bool near = false;
    size_t kj = 0;
    if (j < Na) {
      kj = (size_t)j * Nt + t;
      near = near_pair(C, ki, kj, reach);
      if (near && !EMIT) any_hit |= rect_hit(C, ki, kj, length, width);
    }
    const unsigned long long m = __ballot(near);
    if constexpr (EMIT) {
      if (near) {
        const long long p = offsets[b] + total + __popcll(m & ((1ull << lane) - 1ull));
        pairs[3 * p] = t;
        pairs[3 * p + 1] = i;
        pairs[3 * p + 2] = j;
        // calcEqualInterPlanes (:71-140) with calcPerpendicular (:54-69): bisector planes between {own front, own rear} x
        // {other front, other rear}, offset by rv * distance; agent j gets the negated planes with f2r <-> r2f swapped
        const double Pi[2][2] = {{C.xf[ki], C.yf[ki]}, {C.xr[ki], C.yr[ki]}};
        const double Pj[2][2] = {{C.xf[kj], C.yf[kj]}, {C.xr[kj], C.yr[kj]}};
        double* ci = coef + (size_t)p * 24;
        double* cj = ci + 12;
#pragma unroll
        for (int own = 0; own < 2; ++own)
#pragma unroll
          for (int oth = 0; oth < 2; ++oth) {
            const double x1 = Pi[own][0], y1 = Pi[own][1], x2 = Pj[oth][0], y2 = Pj[oth][1];
            const double a = x2 - x1, bb = y2 - y1;
            const double c = (x1 * x1 + y1 * y1 - x2 * x2 - y2 * y2) / 2;
            const double dx = x1 - x2, dy = y1 - y2;
            const double d = sqrt(dx * dx + dy * dy);
            const double c_i = c + rv * d, c_j = c - rv * d;
            const int slot_i = 2 * own + oth, slot_j = 2 * oth + own;
            ci[3 * slot_i] = a;
            ci[3 * slot_i + 1] = bb;
            ci[3 * slot_i + 2] = c_i;
            cj[3 * slot_j] = -a;
            cj[3 * slot_j + 1] = -bb;
            cj[3 * slot_j + 2] = -c_j;
          }
      }
    }
    total += __popcll(m);
  }
  if constexpr (!EMIT) {
    if (lane == 0) counts[b] = total;
    if (__ballot(any_hit) != 0ull && lane == 0) atomicOr(collide, 1);
  }
}

// exclusive scan of n counts by one 1024-thread workgroup (n = Nt * Na <= a few 100 k): offsets[k], offsets[n] = total
__global__ __launch_bounds__(1024) void k0_scan_kernel(const int* __restrict__ counts, int n, long long* __restrict__ offsets) {
  __shared__ long long part[1024];
  const int tid = (int)threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = tid * per, hi = min(n, lo + per);
  long long s = 0;
  for (int k = lo; k < hi; ++k) s += counts[k];
  part[tid] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const long long v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  long long run = tid ? part[tid - 1] : 0;
  for (int k = lo; k < hi; ++k) {
    offsets[k] = run;
    run += counts[k];
  }
  if (tid == 1023) offsets[n] = part[1023];
}

// ---------------------------------------------------------------------------------------------------------------------
// validator
// ---------------------------------------------------------------------------------------------------------------------
struct Frame {
  double cx, cy, ux, uy;
};
__device__ __forceinline__ Frame frame_of(const double* sol, size_t k, double half_shift) {
  const double yaw = sol[k * 6 + 2];
  Frame f;
  f.ux = cos(yaw);
  f.uy = sin(yaw);
  f.cx = sol[k * 6] + half_shift * f.ux;
  f.cy = sol[k * 6 + 1] + half_shift * f.uy;
  return f;
}

// one lane per (t, pair (i < j)): separating axes of both rectangles, touching counts as overlap
__global__ void validate_vehicles_kernel(const double* __restrict__ sol, int Na, int Nt, double half_shift, double hl, double hw,
                                         unsigned long long* __restrict__ out /* [0] count, [1] first (t, i, j) packed */) {
  const long long n_pairs = (long long)Na * (Na - 1) / 2;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_pairs * Nt) return;
  const int t = (int)(g / n_pairs);
  long long q = g - (long long)t * n_pairs;
  // row i of the strict upper triangle that holds q
  int i = (int)(((2.0 * Na - 1.0) - sqrt((2.0 * Na - 1.0) * (2.0 * Na - 1.0) - 8.0 * (double)q)) * 0.5);
  while ((long long)i * (2 * Na - i - 1) / 2 > q) --i;
  while ((long long)(i + 1) * (2 * Na - i - 2) / 2 <= q) ++i;
  const int j = (int)(q - (long long)i * (2 * Na - i - 1) / 2) + i + 1;
  const Frame A = frame_of(sol, (size_t)i * Nt + t, half_shift), B = frame_of(sol, (size_t)j * Nt + t, half_shift);
  const double dx = B.cx - A.cx, dy = B.cy - A.cy;
  const double anx = -A.uy, any_ = A.ux, bnx = -B.uy, bny = B.ux;
  auto sep = [&](double ux, double uy, double nx, double ny, double oux, double ouy, double onx, double ony) {
    const double eu = hl * fabs(oux * ux + ouy * uy) + hw * fabs(onx * ux + ony * uy) + hl;
    const double en = hl * fabs(oux * nx + ouy * ny) + hw * fabs(onx * nx + ony * ny) + hw;
    return (fabs(dx * ux + dy * uy) <= eu) && (fabs(dx * nx + dy * ny) <= en);
  };
  if (sep(A.ux, A.uy, anx, any_, B.ux, B.uy, bnx, bny) && sep(B.ux, B.uy, bnx, bny, A.ux, A.uy, anx, any_)) {
    atomicAdd(&out[0], 1ull);
    atomicMin(&out[1], ((unsigned long long)t << 40) | ((unsigned long long)i << 20) | (unsigned long long)j);
  }
}

// one lane per (t, agent, obstacle): signed distance rectangle - disc; also the out-of-map test per (t, agent) on o == 0
__global__ void validate_obstacles_kernel(const double* __restrict__ sol, int Na, int Nt, const double* __restrict__ obs, int n_obs,
                                          double half_shift, double hl, double hw, double dimx, double dimy, int check_map,
                                          unsigned long long* __restrict__ out /* [2] count, [3] first, [4] out of map, [5] min clearance bits */) {
  const int n_o = n_obs > 0 ? n_obs : 1;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)Na * Nt * n_o) return;
  const int o = (int)(g % n_o);
  const long long at = g / n_o;
  const int t = (int)(at % Nt), a = (int)(at / Nt);
  const Frame F = frame_of(sol, (size_t)a * Nt + t, half_shift);
  const double nx = -F.uy, ny = F.ux;
  if (n_obs > 0) {
    const double rx = obs[3 * o] - F.cx, ry = obs[3 * o + 1] - F.cy;
    const double lx = fabs(rx * F.ux + ry * F.uy) - hl, ly = fabs(rx * nx + ry * ny) - hw;
    const double dist = hypot(fmax(lx, 0.0), fmax(ly, 0.0)) + fmin(fmax(lx, ly), 0.0) - obs[3 * o + 2];
    if (dist < 0.0) {
      atomicAdd(&out[2], 1ull);
      atomicMin(&out[3], ((unsigned long long)t << 40) | ((unsigned long long)a << 20) | (unsigned long long)o);
    }
    // min over doubles through an order-preserving integer key
    long long bits = __double_as_longlong(dist);
    const unsigned long long key = bits < 0 ? ~(unsigned long long)bits : ((unsigned long long)bits | 0x8000000000000000ull);
    atomicMin(&out[5], key);
  }
  if (check_map && o == 0) {
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double sx = (c & 1) ? 1.0 : -1.0, sy = (c & 2) ? 1.0 : -1.0;
      const double px = F.cx + sx * hl * F.ux + sy * hw * nx, py = F.cy + sx * hl * F.uy + sy * hw * ny;
      bad |= (px < 0.0) || (px > dimx) || (py < 0.0) || (py > dimy);
    }
    if (bad) atomicAdd(&out[4], 1ull);
  }
}

// one lane per (agent, frame): getState of scripts/visualize.py:256-281, the same operations in the same order
__global__ void expand_frames_kernel(const double* __restrict__ sol, int Na, int Nt, int S, double* __restrict__ frames) {
  const int nf = (Nt - 1) * S + 1;
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= (long long)Na * nf) return;
  const int a = (int)(g / nf), f = (int)(g - (long long)a * nf);
  const double t = (double)f / (double)S;
  const int idx = (int)ceil(t);                       // first state whose time is >= t
  double* o = frames + (size_t)g * 6;
  const double* nx = sol + ((size_t)a * Nt + idx) * 6;
  double px = nx[0], py = nx[1], pyaw = nx[2];
  if (idx > 0) {
    const double* ls = nx - 6;
    double yaw_last = ls[2];
    const double yaw_next = nx[2];
    const double pi = 3.141592653589793;
    if ((yaw_last - yaw_next) > pi) yaw_last = yaw_last - 2 * pi;
    else if ((yaw_next - yaw_last) > pi) yaw_last = yaw_last + 2 * pi;
    const double tau = (t - (double)(idx - 1)) / 1.0;
    px = (nx[0] - ls[0]) * tau + ls[0];
    py = (nx[1] - ls[1]) * tau + ls[1];
    pyaw = (yaw_next - yaw_last) * tau + yaw_last;
  }
  o[0] = px;
  o[1] = py;
  o[2] = pyaw;
  o[3] = o[4] = o[5] = 0.0;
}

}  // namespace

hipError_t expand_frames_launch(const double* sol, int Na, int Nt, int S, double* frames, hipStream_t s) {
  const long long n = (long long)Na * ((long long)(Nt - 1) * S + 1);
  hipLaunchKernelGGL(expand_frames_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, sol, Na, Nt, S, frames);
  return hipGetLastError();
}

hipError_t k0_count(const K0Centres& c, int Na, int Nt, double reach, float length, float width, int* counts, int* collide,
                    long long* offsets, hipStream_t s) {
  CentresDev C{c.xf, c.yf, c.xr, c.yr, c.xc, c.yc, c.cs, c.sn};
  hipLaunchKernelGGL(k0_pairs_kernel<false>, dim3(Na * Nt), dim3(64), 0, s, C, Na, Nt, reach, length, width, 0.0, counts,
                     collide, (const long long*)nullptr, (int32_t*)nullptr, (double*)nullptr);
  hipLaunchKernelGGL(k0_scan_kernel, dim3(1), dim3(1024), 0, s, (const int*)counts, Na * Nt, offsets);
  return hipGetLastError();
}

hipError_t k0_emit(const K0Centres& c, int Na, int Nt, double reach, float length, float width, double rv,
                   const long long* offsets, int32_t* pairs, double* coef, hipStream_t s) {
  CentresDev C{c.xf, c.yf, c.xr, c.yr, c.xc, c.yc, c.cs, c.sn};
  hipLaunchKernelGGL(k0_pairs_kernel<true>, dim3(Na * Nt), dim3(64), 0, s, C, Na, Nt, reach, length, width, rv,
                     (int*)nullptr, (int*)nullptr, offsets, pairs, coef);
  return hipGetLastError();
}

hipError_t validate_launch(const double* sol, int Na, int Nt, const double* obs, int n_obs, double half_shift, double hl,
                           double hw, double dimx, double dimy, int check_map, unsigned long long* out, hipStream_t s) {
  const long long n1 = (long long)Na * (Na - 1) / 2 * Nt;
  if (n1 > 0)
    hipLaunchKernelGGL(validate_vehicles_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, sol, Na, Nt, half_shift,
                       hl, hw, out);
  const long long n2 = (long long)Na * Nt * (n_obs > 0 ? n_obs : 1);
  if (n_obs > 0 || check_map)
    hipLaunchKernelGGL(validate_obstacles_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, s, sol, Na, Nt, obs, n_obs,
                       half_shift, hl, hw, dimx, dimy, check_map, out);
  return hipGetLastError();
}

}  // namespace csdo
