// dsqp_layout.h — the constants that size an agent's LDS working set: how much of a node's factor stays in registers, the leading
// dimensions of the lane-major LDS arrays, the capacity of the dense BCR tail.  Plain constants (no device code): the program
// (dsqp_program.h) and the launcher's / the lane-serial build's class rule (dsqp_class.h) share them.
#pragma once

namespace csdo {

// BCR stops when at most TAIL_NODES nodes remain (see dsqp_program.h)
constexpr int TAIL_NODES = 6, TAIL_N = 6 * TAIL_NODES;
// The 512-thread class may keep a larger tail where that saves a level of the reduction and the agent's LDS has room for the
// inverse (AgentDesc::tail_nodes = 8 or 12, dsqp_class.h): its kernels read the tail's size at run time (BIGT below).
constexpr int TAIL_NODES_BIG = 12, TAIL_N_BIG = 6 * TAIL_NODES_BIG;
#if !defined(CSDO_TAIL_BIG)
#define CSDO_TAIL_BIG 1   // 0: six nodes for every agent; 1: eight where that saves a level (horizons 193 .. 256); 2: also a FOLDED tail of
                          // twelve (horizons 97 .. 192 of the 512-thread class): six nodes inverted densely, the inverse expanded by the last level's factors
#endif
// (2, round 6: measured, not shipped - per iteration a folded agent saves a level in both sweeps, 1.6 k cycles of 14 k, and gives half
//  of it back waiting for the 72 x 72 tail product, whose 72 LDS reads per lane on four row waves saturate the LDS pipe; the expansion
//  costs 20 k cycles per SQP iteration and the larger inverse takes the LDS that cached the planes' coefficients: map100 53.2 -> 54.0 ms,
//  one instance alone 8.32 -> 8.19 ms; profiles/r06_phase_profile_map100_folded_tail.txt)

#if !defined(CSDO_ER_REG)
#define CSDO_ER_REG 34   // round 5 (after the spills of the cold phases went: no reload in the levels at 32, 34, 36 any more), map100 / synth1024 ms per step: 30: 56.90 / 38.14, 32: 56.70 / 37.95, 34: 55.99 / 37.29, 36: 56.39 / 37.59; pair-split solve of round 4: 12: 73.4, 18: 71.5, 24: 66.4, 30: 65.8, 36: 71.9 (32 and 34 put scratch reloads into the levels then); one-lane form of round 3: 20: 83.1, 24: 79.9, 32: 80.2, 36: 81.0
#endif
#if !defined(CSDO_ER_REG1)
#define CSDO_ER_REG1 30   // the same for residency mode 1 (its solve still has scratch reloads: with 34 the mode-1 agents of the room set went from 27 to 32 us per iteration)
#endif
constexpr int ER_REG = CSDO_ER_REG, FX_ER = 36 - ER_REG;   // fx[lane] = F_r[ER_REG..36) then the packed pivot inverse
constexpr int ER_REG1 = CSDO_ER_REG1, FX_ER1 = 36 - ER_REG1;
#if !defined(CSDO_ER_REG2)
#define CSDO_ER_REG2 0
#endif
constexpr int ER_REG2 = CSDO_ER_REG2, FX_ER2 = 36 - ER_REG2;


// lane-major leading dimensions (doubles per lane) of the LDS arrays.  All are 2 * odd: 16-byte aligned lanes, and the
// ds_read_b128 / ds_write_b128 of 16 consecutive lanes (also of lanes a power of two apart) fall into 16 different
// 4-bank groups - conflict free.  (12 doubles, the former reduction stride, is 2-way conflicting.)
constexpr int LD_vec = 6, LD_pl = 6, LD_pr = 6, LD_rhs = 6, LD_carry = 6, LD_carry2 = 6, LD_red = 14, LD_lohi = 22, LD_fx = 2 * (((FX_ER + 21 + 1) / 2) | 1),
              // the block's per-timestep arrays (vec .. fx) double as the factorisation's exchange columns, 78 fields x stride
              LD_block = (24 + LD_lohi + LD_fx) > 78 ? (24 + LD_lohi + LD_fx) : 78,
              LD_fx2 = 34, LD_block2 = 18 + LD_fx2,   // mode 2: vec, pr (= rhs), carry (whose two spare doubles per lane hold the block's last two entries) + fx
              LD_fx1 = 2 * (((FX_ER1 + 1) / 2) | 1), LD_block1 = 24 + LD_lohi + LD_fx1,   // mode 1: mode 0 without the pivot inverse (and the rows' state)
              LD_stash = 38, LD_tinv = 38, LD_prow = 10;

}  // namespace csdo
