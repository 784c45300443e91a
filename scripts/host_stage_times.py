"""Diagnostic (host code only; run it where the timing matters - the GPU box has 256 cores, the build container 8): the DO phase's host
stages on the library's thread pool (csrc/host_pool.h).  One world's bridge call after call (the first call starts the pool), the
batched bridge and the packing (csdo_dsqp_estimate_work = pack_worlds) for 1 / 5 / 16 / 60 worlds, with CSDO_HOST_THREADS as set.
usage: python scripts/host_stage_times.py [map100] [n_worlds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csdotrajectoryplanning_amd import workloads  # noqa: E402
from csdotrajectoryplanning_amd.solver import estimate_work, interpolate_and_planes, interpolate_and_planes_batch_host  # noqa: E402

if __name__ == "__main__":
    wl = sys.argv[1] if len(sys.argv) > 1 else "map100"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    built = [workloads.build_job(j) for j in workloads.workload_jobs(wl, n)]
    items = [(*info["paths"], w.dimx, w.dimy, w.obstacles) for w, info in built]
    w0 = built[0][0]
    print("host threads: cpu_count %d, CSDO_HOST_THREADS=%s" % (os.cpu_count(), os.environ.get("CSDO_HOST_THREADS")))
    st, ac, po, G = built[0][1]["paths"]
    one = []
    for k in range(12):
        if k == 6:
            time.sleep(0.05)                       # the pool's threads are asleep again
        t = time.perf_counter()
        interpolate_and_planes(st, ac, po, G, w0.veh, w0.parm, w0.dimx, w0.dimy, w0.obstacles)
        one.append((time.perf_counter() - t) * 1e3)
    print("one world's bridge (csdo_preprocess), ms, call after call (a 50 ms pause before the seventh):", " ".join("%.2f" % v for v in one))
    for m in (1, 5, 16, len(items)):
        if m > len(items):
            continue
        tb, tp = [], []
        for _ in range(6):
            t = time.perf_counter()
            b = interpolate_and_planes_batch_host(items[:m], w0.veh, w0.parm)
            tb.append((time.perf_counter() - t) * 1e3)
            worlds = [x[0] for x in b]
            t = time.perf_counter()
            estimate_work(worlds)
            tp.append((time.perf_counter() - t) * 1e3)
        print("%2d worlds: bridge (csdo_preprocess_batch) best %.2f ms, median %.2f; packing (csdo_dsqp_estimate_work) best %.2f ms, median %.2f" % (
            m, min(tb), sorted(tb)[len(tb) // 2], min(tp), sorted(tp)[len(tp) // 2]))
