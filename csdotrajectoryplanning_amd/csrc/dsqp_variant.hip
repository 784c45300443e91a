// One instantiation of dsqp_agent_kernel per translation unit (see Makefile: VARIANTS).  CSDO_V_SPLIT: bit 0 = two specialised lanes per
// timestep (always), bits 1.. = REFINE (csdo_qp_parm::solve_refinement: 1 -> 3, 2 -> 5).
#include "dsqp_kernel_body.h"
namespace csdo {
template hipError_t launch_variant<CSDO_V_BLOCK, CSDO_V_MODE, ((CSDO_V_SPLIT & 1) != 0), (CSDO_V_SPLIT >> 1)>(const DeviceBatch&, const LaunchGroup&, int,
                                                                                                              hipStream_t);
}
