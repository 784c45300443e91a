// Launch interface between the C ABI (capi.hip) and the kernels (dsqp_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "csdo_device_types.h"

namespace csdo {
size_t dsqp_lds_bytes(int max_nt, int max_obs, int max_planes, bool big);
hipError_t launch_dsqp(const DeviceBatch& B, int max_nt, int max_obs, int max_planes, hipStream_t stream);
hipError_t launch_boxes(const double* pts, int n, const double* obs, int n_obs, double dimx, double dimy, double rv,
                        double* boxes, int* status, hipStream_t stream);
}  // namespace csdo
