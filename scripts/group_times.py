"""Per launch group of one batch: how many agents, the sum of their device times over the 256 CUs, the longest, the median.
usage (GPU box): python scripts/group_times.py [workload]        (CSDO_DIAG_LIB picks the library)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from csdotrajectoryplanning_amd import workloads  # noqa: E402
from csdotrajectoryplanning_amd.solver import DsqpHandle  # noqa: E402


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "room50"
    worlds = [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs(wl), 16)]
    h = DsqpHandle(0)
    h.upload(worlds)
    h.run()
    h.run()
    sols = h.download()
    secs = np.concatenate([s.agent_seconds for s in sols]) * 1e3
    admm = np.concatenate([s.admm_iters for s in sols])
    nt = np.concatenate([np.full(w.Na, w.Nt) for w in worlds])
    grp = np.asarray(h.agent_groups())
    for g, info in enumerate(h.launch_groups()):
        m = grp == g
        print("group %d: %d threads mode %d, %d agents, kernel %.1f ms, sum/256 %.1f ms, longest %.1f, median %.1f, us/iter median %.1f (Nt %d..%d)" % (
            g, info["threads"], info["residency_mode"], m.sum(), info["seconds"] * 1e3, secs[m].sum() / 256, secs[m].max(),
            np.median(secs[m]), np.median(secs[m] * 1e3 / np.maximum(admm[m], 1)), nt[m].min(), nt[m].max()))
    top = np.argsort(-secs)[:6]
    print("longest agents:", [(int(nt[a]), int(admm[a]), round(float(secs[a]), 1), int(grp[a])) for a in top])
    print("all: sum/256 %.1f ms" % (secs.sum() / 256))


if __name__ == "__main__":
    main()
