// One instantiation of dsqp_agent_kernel per translation unit (see Makefile: VARIANTS).  CSDO_V_SPLIT: bit 0 = two specialised lanes per
// timestep (always), bit 1 = REFINE (csdo_qp_parm::solve_refinement).
#include "dsqp_kernel_body.h"
namespace csdo {
template hipError_t launch_variant<CSDO_V_BLOCK, CSDO_V_MODE, ((CSDO_V_SPLIT & 1) != 0), ((CSDO_V_SPLIT & 2) != 0)>(const DeviceBatch&, const LaunchGroup&,
                                                                                                                   int, hipStream_t);
}
