// Host bridge of the DO backend: coarse front-end paths -> fixed-length initial guess, neighbour pairs and
// separating half-planes.  Replaces InterpolateInitalGuess / findNeighborPairsByTrustRegion / calcEqualInterPlanes
// (sqp/inter_agent_cons.cc:12-140,143-411 of the reference; call sites csdo.cc:116-129).
// O(Na*Nt) interpolation and O(Nt*Na^2) pair search on precomputed float disc centres: host work by design
// (SURVEY 8a rows a2-a4), it is ~1 ms at 50 agents.
#pragma once
#include <cstddef>
#include <vector>

#include "../../include/csdo_dsqp.h"

namespace csdo {
// float disc / rectangle centres and heading cosines of every (agent, t), as the reference's State constructor stores
// them (common/motion_planning.h:115-132): the inputs of the neighbour search and of the plane generation
struct BridgeCentres {
  int Na = 0, Nt = 0;
  std::vector<float> xf, yf, xr, yr, xc, yc, cs, sn;   // [Na][Nt]
};
// stages of bridge_preprocess (the device path, csdo_preprocess_device, replaces the two O(Nt Na^2) ones)
int bridge_interpolate(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                       const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* out,
                       BridgeCentres& C);
bool bridge_pairs(const BridgeCentres& C, double r_trust, const csdo_vehicle* veh, std::vector<int32_t>& pairs);
int bridge_planes(const BridgeCentres& C, const std::vector<int32_t>& pairs, const csdo_vehicle* veh, const double* coef,
                  csdo_bridge_out* out);
int bridge_planes(const BridgeCentres& C, const int32_t* pairs, size_t n_pairs, const csdo_vehicle* veh, const double* coef,
                  csdo_bridge_out* out);   // the same on a slice of a batch's pair list
int bridge_preprocess(const double* states, const int32_t* actions, const int32_t* path_off, int32_t Na,
                      const double* goals, const csdo_vehicle* veh, const csdo_qp_parm* parm, csdo_bridge_out* out);
void bridge_free(csdo_bridge_out* out);
}  // namespace csdo
