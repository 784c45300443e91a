// Launch interface between the C ABI (capi.hip) and the kernels (dsqp_kernel.hip).
#pragma once
#include <hip/hip_runtime.h>

#include "csdo_device_types.h"
#include "dsqp_class.h"

namespace csdo {
// One launch = one group of agents that share a kernel instantiation: workgroup size by horizon and LDS residency
// mode by working set (agent_program in dsqp_program.h: 0 everything in LDS ... 2 only the 6-vectors).
struct LaunchGroup {
  int first = 0, count = 0;   // range in DeviceBatch::order
  int block = 0;              // threads per workgroup: 256 (Nt <= 128), 512 (<= 256), 1024 (<= 512)
  int mode = 0;
  int max_nt = 0;
  size_t lds_bytes = 0;       // dynamic LDS of the launch: the largest working set among the group's agents
  double seconds = 0.0;       // duration of the last launch (HIP events on the group's stream)
  int* queue = nullptr;       // device counter of the group's agent queue (persistent workgroups)
  int primary = 0, elastic = 0;   // workgroups of the first launch (the group's share of the CUs) and of the second one
};
int dsqp_workgroups_per_cu(int block, size_t lds_bytes);            // persistent workgroups of a launch group one CU holds
size_t dsqp_lds_capacity_two_per_cu();                             // ... of the class that runs two workgroups per CU
size_t dsqp_lds_capacity();                                        // dynamic LDS one workgroup may ask for
// Launches `workgroups` persistent workgroups that drain the group's queue (g.queue must have been zeroed on a stream
// this launch is ordered after); several launches may share one queue.
hipError_t launch_dsqp(const DeviceBatch& B, const LaunchGroup& g, int workgroups, hipStream_t stream);
hipError_t launch_boxes(const double* pts, int n, const double* obs, int n_obs, double dimx, double dimy, double rv,
                        double* boxes, int* status, hipStream_t stream);
hipError_t launch_math_probe(int fn, const double* a, const double* b, double* out, int n, hipStream_t stream);
}  // namespace csdo
