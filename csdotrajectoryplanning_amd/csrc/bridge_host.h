// Host bridge of the DO backend: coarse front-end paths -> fixed-length initial guess, neighbour pairs and
// separating half-planes.  Replaces InterpolateInitalGuess / findNeighborPairsByTrustRegion / calcEqualInterPlanes
// (sqp/inter_agent_cons.cc:12-140,143-411 of the reference; call sites csdo.cc:116-129).
// O(Na*Nt) interpolation and O(Nt*Na^2) pair search on precomputed float disc centres: host work by design
// (SURVEY 8a rows a2-a4), it is ~1 ms at 50 agents.
#pragma once
#include "../../include/csdo_dsqp.h"

namespace csdo {
int bridge_preprocess(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                      const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* out);
void bridge_free(csdo_bridge_out* out);
}  // namespace csdo
