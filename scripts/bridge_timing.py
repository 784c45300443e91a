import sys, os, time
sys.path.insert(0,'/root/repo')
import torch
from csdotrajectoryplanning_amd import workloads
from csdotrajectoryplanning_amd.solver import DsqpHandle
if __name__=="__main__":
    built=workloads.build_jobs_parallel(workloads.workload_jobs("map100"),16)
    items=[(*i["paths"], w.dimx,w.dimy,w.obstacles) for w,i in built]
    h=DsqpHandle(0); w0=built[0][0]
    for k in range(3):
        t=time.perf_counter(); r=h.interpolate_and_planes_batch(items,w0.veh,w0.parm); print("call %.2f ms"%((time.perf_counter()-t)*1e3), file=sys.stderr)
