// One instantiation of dsqp_agent_kernel per translation unit (see Makefile: VARIANTS).
#include "dsqp_kernel_body.h"
namespace csdo {
template hipError_t launch_variant<CSDO_V_BLOCK, CSDO_V_MODE, (CSDO_V_SPLIT != 0)>(const DeviceBatch&, const LaunchGroup&, int,
                                                                                       hipStream_t);
}
