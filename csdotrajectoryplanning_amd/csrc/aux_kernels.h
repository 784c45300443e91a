// Launch interface of aux_kernels.hip (device neighbour search / planes, trajectory validator).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace csdo {
struct K0Centres {   // device pointers, [Na][Nt] floats each (BridgeCentres of bridge_host.h)
  const float *xf, *yf, *xr, *yr, *xc, *yc, *cs, *sn;
};
// pass 1: counts[Na*Nt] pairs per (t, i), collide flag, then offsets[Na*Nt + 1] (exclusive scan, last = number of pairs)
hipError_t k0_count(const K0Centres& c, int Na, int Nt, double reach, float length, float width, int* counts, int* collide,
                    long long* offsets, hipStream_t s);
// pass 2: pairs[n][3] in (t, i, j) order and coef[n][24] (agent i's 12 plane coefficients, then agent j's)
hipError_t k0_emit(const K0Centres& c, int Na, int Nt, double reach, float length, float width, double rv,
                   const long long* offsets, int32_t* pairs, double* coef, hipStream_t s);
// out[6] (zero / ~0 initialised by the caller): vehicle hits, first (t<<40|i<<20|j), obstacle hits, first (t<<40|a<<20|o),
// out-of-map (t, agent) count, minimum clearance as an order-preserving key
hipError_t validate_launch(const double* sol, int Na, int Nt, const double* obs, int n_obs, double half_shift, double hl,
                           double hw, double dimx, double dimy, int check_map, unsigned long long* out, hipStream_t s);
// frames[Na][(Nt - 1) * S + 1][6] = the poses the authors' animation looks at with S frames per move (getState,
// scripts/visualize.py:256-281: linear in x, y, yaw between states, the earlier yaw moved by 2 pi when they are more than pi apart)
hipError_t expand_frames_launch(const double* sol, int Na, int Nt, int S, double* frames, hipStream_t s);
}  // namespace csdo
