// dsqp_class.h — kernel class of one agent: workgroup size, LDS residency mode, whether its inter-vehicle rows' state fits LDS,
// and the capacity of its dense BCR tail.  Plain C++ (no HIP): the launcher (dsqp_kernel.hip, capi.hip) and the lane-serial test
// build (tests/emu) share it, because the tail's size decides the elimination order - the one item of the class that the
// results' last bits depend on.  Every item is a function of the agent alone (horizon, obstacles of its world, its planes).
#pragma once
#include <cstddef>

#include "dsqp_layout.h"

namespace csdo {

constexpr size_t LDS_CAP = 160 * 1024 - 64;      // 160 KB per workgroup minus the kernel's static LDS (queue slot)
constexpr size_t LDS_CAP_2WG = 80 * 1024 - 64;   // two workgroups of the 256-thread class per CU

// LDS working set of one agent in bytes
inline size_t dsqp_lds_bytes(int nt, int n_obs, int n_planes, int mode, bool rows_lds, int tail_nodes = TAIL_NODES) {
  const int st = (nt + 1) & ~1;
  // exchange vectors vec, pr, rhs, carry (6 each) + bounds of the home rows 22 (the t -> t-1 hand-over aliases them) +
  // the third of the factor that is not in the solver lane's registers 34 (mode 0); mode 3: vec, pr, rhs, carry, carry2
  const size_t per_lane = mode == 3 ? 30 : (mode == 2 ? (size_t)LD_block2 : (mode == 1 ? (size_t)LD_block1 : (size_t)LD_block));
  const size_t n_obs_pad = (3 * (size_t)n_obs + 1) & ~(size_t)1, n_pc_pad = (3 * (size_t)n_planes + 1) & ~(size_t)1;
  const size_t planes = (mode == 0 && rows_lds) ? n_pc_pad + (size_t)LD_prow * n_planes : 0;   // rhs shares + duals / slacks
  // the tail: gathered rhs + hand-over sums, and the explicit inverse (rows of 6 tn + 2)
  const size_t tail = tail_nodes <= TAIL_NODES ? (size_t)(2 * TAIL_N + TAIL_N * 38)
                                               : (size_t)(2 * TAIL_N_BIG + 6 * tail_nodes * (6 * tail_nodes + 2));
  return (per_lane * st + n_obs_pad + 32 + tail + planes) * sizeof(double);
}

// stride of the last reduction level for a tail of at most `tail_nodes` nodes (agent_program's h_tail)
inline int dsqp_tail_stride(int nt, int tail_nodes) {
  int h = 1;
  while ((nt + h - 1) / h > tail_nodes) h <<= 1;
  return h;
}

// (CSDO_TAIL_BIG: dsqp_layout.h)
// returns the workgroup size; sets the residency mode, whether the rows' state fits LDS, and the tail's capacity
inline int dsqp_agent_class(int nt, int n_obs, int n_planes, int* mode, int* rows_lds, int* tail_nodes) {
  // workgroup size: two specialised lanes per timestep (Nt <= 128: 256 threads, <= 256: 512 threads, <= 512: 1024
  // threads with 128 registers per lane: correct but spills; horizons that long are outside the benchmark sets)
  // The 256-thread class runs two workgroups per CU, so it only takes agents whose working set fits half the LDS; a
  // short horizon that does not (Nt > ~105 with 25 obstacles) runs in the 512-thread class with half its lanes idle.
  // Horizons 257 .. 384 take 768 threads: three waves per SIMD leave 168 registers per lane instead of 128 (measured on the
  // room set, whose long agents have 257 .. 295 timesteps).
  int block = nt <= 128 ? 256 : (nt <= 256 ? 512 : (nt <= 384 ? 768 : 1024));
  if (block == 256 && dsqp_lds_bytes(nt, n_obs, n_planes, 0, false) > LDS_CAP_2WG) block = 512;
  *rows_lds = 0;
  *tail_nodes = TAIL_NODES;
  if (block >= 768) {   // nothing of the factor in registers: F_r in LDS where that fits (mode 2), else from the workspace
    *mode = (block == 768 && dsqp_lds_bytes(nt, n_obs, n_planes, 2, false) <= LDS_CAP) ? 2 : 3;
  } else if (dsqp_lds_bytes(nt, n_obs, n_planes, 0, true) <= (block == 256 ? LDS_CAP_2WG : LDS_CAP)) {   // (256: keep two per CU)
    *mode = 0;
    *rows_lds = 1;
  } else {
    *mode = dsqp_lds_bytes(nt, n_obs, n_planes, 0, false) <= LDS_CAP ? 0 : 1;
    // An obstacle list that does not even fit beside the lean 512-thread layout (mode 1: 52 doubles per timestep) runs in the
    // 768-thread class, whose modes keep 52 / 30 doubles per timestep in LDS - with lanes to spare for a horizon this short,
    // slower, but it runs: 5000 obstacles beside 100 timesteps, 3900 beside 200 (mode 1's growth from 46 to 52
    // doubles had turned worlds away that round 3 accepted; the reference has no such limit at all).
    if (*mode == 1 && dsqp_lds_bytes(nt, n_obs, n_planes, 1, false) > LDS_CAP) {
      block = 768;
      *mode = dsqp_lds_bytes(nt, n_obs, n_planes, 2, false) <= LDS_CAP ? 2 : 3;
    }
  }
  // A larger dense tail for the 512-thread class: the explicit inverse of up to 8 (12) nodes instead of 6 where that makes the
  // reduction one level shorter - a level is a tenth of an ADMM iteration - and the inverse fits beside everything the class
  // decided above (same mode, the rows' state where it was).
  if (CSDO_TAIL_BIG && block == 512) {
    const int h6 = dsqp_tail_stride(nt, TAIL_NODES);
    const int cands[2] = {8, TAIL_NODES_BIG};
    for (int c = 0; c < (CSDO_TAIL_BIG >= 2 ? 2 : 1); ++c) {
      const int tn = cands[c];
      if (dsqp_tail_stride(nt, tn) < h6 && dsqp_lds_bytes(nt, n_obs, n_planes, *mode, *rows_lds != 0, tn) <= LDS_CAP) {
        *tail_nodes = tn;
        break;
      }
    }
  }
  return block;
}

}  // namespace csdo
