// front_end.h — host front end: PBS over spatiotemporal hybrid A* (see front_end.cc).
#pragma once
#include <vector>

#include "../../include/csdo_dsqp.h"

namespace csdo {
int front_end_plan(const double* starts, const double* goals, int Na, double dimx, double dimy, const double* obstacles,
                   int n_obs, const csdo_vehicle* veh, const csdo_front_end_parm* parm, csdo_paths* out);
void front_end_free(csdo_paths* p);
}  // namespace csdo
