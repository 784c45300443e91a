import os, sys, json
sys.path.insert(0, '.')
import torch
from csdotrajectoryplanning_amd import workloads
from csdotrajectoryplanning_amd.solver import DsqpHandle
out = {}
for name, k in (("map100", 0), ("map100", 3), ("map50", 0)):
    w, _ = workloads.build_job(workloads.workload_jobs(name, k + 1)[k])
    h = DsqpHandle(0); h.upload([w]); h.run()
    ks = min(h.run() for _ in range(5)); s = h.download()[0]
    out["%s_ex%d" % (name, k)] = (round(ks * 1e3, 3), int(s.admm_iters.sum()))
    h.close()
print(os.environ.get("CSDO_DIAG_LIB"), out)
