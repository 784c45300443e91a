"""Diagnostic (GPU box): the PCIe-inclusive DO phase of every workload, best of 5 with 50 ms of sleep in front of each call -
csdo_do_phase (one library call: DsqpHandle.do_phase) against the same pipeline driven from Python (DsqpHandle.do_phase_stream,
with the host threads' bridge and, for one-launch jobs, with round 4's device bridge).
usage: python scripts/do_phase_times.py [workload ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from csdotrajectoryplanning_amd import workloads  # noqa: E402
from csdotrajectoryplanning_amd.solver import DsqpHandle  # noqa: E402

for wl in (sys.argv[1:] or ["map100", "map50", "room50", "agents100"]):
    built = [workloads.build_job(j) for j in workloads.workload_jobs(wl)]
    items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built if w.Na == len(info["paths"][2]) - 1]
    if len(items) != len(built):
        print(wl, "holds partial worlds: skipped")
        continue
    w0 = built[0][0]
    h = DsqpHandle(0)
    h.upload([w for w, _ in built])
    h.run()
    resident = min(h.run() for _ in range(3)) * 1e3
    forms = [("csdo_do_phase (one call)", lambda out: h.do_phase(items, w0.veh, w0.parm, out=out)[:2]),
             ("python, host-thread bridge", lambda out: h.do_phase_stream(items, w0.veh, w0.parm, out=out)),
             ("python, device bridge", lambda out: h.do_phase_stream(items, w0.veh, w0.parm, out=out, device_bridge=True))]
    for name, call in forms:
        out, best = None, None
        for _ in range(6):
            time.sleep(0.05)
            out, tm = call(out)
            if best is None or tm["total"] < best["total"]:
                best = tm
        print("%-10s %-28s total %.1f ms = kernels %.1f + %.1f  first launch %.2f  kernels done %.1f  chunks %s" % (
            wl, name, best["total"] * 1e3, resident, best["total"] * 1e3 - resident, best["first_launch"] * 1e3, best["kernels_done"] * 1e3,
            [(c["worlds"], round(c["bridge"] * 1e3, 2), round(c["upload"] * 1e3, 2), round(c["kernel"] * 1e3, 1)) for c in best["chunks"]]))
    h.close()
