"""Diagnostic: the streamed DO phase of a workload (DsqpHandle.do_phase_stream) for several chunkings, best of 4 each.
usage (GPU box): python scripts/stream_fractions.py [map100] ["0.08,0.27,0.65;0.05,0.2,0.75;..."] [pause_ms]
pause_ms: sleep that long in front of every DO phase (a planner calls the DO phase once per search: the library's host threads are
asleep and the clocks are down when it starts - the back-to-back calls of a benchmark loop never see that)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (first: the HIP runtime of the process)
from csdotrajectoryplanning_amd import workloads  # noqa: E402
from csdotrajectoryplanning_amd.solver import DsqpHandle  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "map100"
sets = sys.argv[2] if len(sys.argv) > 2 else "0.08,0.27,0.65;0.05,0.2,0.75;0.04,0.16,0.8;0.03,0.12,0.25,0.6;0.05,0.95;0.1,0.3,0.6"
pause = float(sys.argv[3]) * 1e-3 if len(sys.argv) > 3 else 0.0
built = [workloads.build_job(j) for j in workloads.workload_jobs(wl)]
items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built]
w0 = built[0][0]
h = DsqpHandle(0)
h.upload([w for w, _ in built])
h.run()
print("resident batch: kernels %.1f ms; pause in front of every DO phase %.0f ms" % (min(h.run() for _ in range(3)) * 1e3, pause * 1e3))
for fr in sets.split(";"):
    hr_set = (True,)
    if fr.endswith("!"):          # "...!": with the results in device memory too (csdo_dsqp_set_host_results off)
        fr, hr_set = fr[:-1], (False, True)
    f = tuple(float(x) for x in fr.split(","))
    for hr in hr_set:
        out, best = None, None
        for _ in range(6):
            if pause:
                time.sleep(pause)
            out, tm = h.do_phase_stream(items, w0.veh, w0.parm, fractions=f, out=out, min_first_agents=0, host_results=hr)
            if best is None or tm["total"] < best["total"]:
                best = tm
        print("%-24s %s total %.1f ms  first launch %.1f  kernels done %.1f  chunks %s" % (
            fr, "results -> host memory  " if hr else "results -> device memory", best["total"] * 1e3, best["first_launch"] * 1e3, best["kernels_done"] * 1e3,
            [(c["worlds"], round(c["kernel"] * 1e3, 1)) for c in best["chunks"]]))
        print("   bridge / upload of the chunks, ms:", [(round(c["bridge"] * 1e3, 2), round(c["upload"] * 1e3, 2), {k: round(v * 1e3, 2) for k, v in c.get("upload_parts", {}).items() if k in ("pack", "stage", "h2d")}) for c in best["chunks"]])
h.close()
