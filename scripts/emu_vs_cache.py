"""Bit-level regression of the lane-serial build against the cached whole-workload results of scripts/arbiter_run.py ("product",
"product_refined"): a source change that must not change a bit (hygiene, refactoring) is checked in a minute, without a GPU.
   python scripts/emu_vs_cache.py [workloads, comma separated] [--mode 0]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))

if __name__ == "__main__":
    import arbiter_run
    from csdotrajectoryplanning_amd import workloads
    from tests import emu_lib
    names = (sys.argv[1] if len(sys.argv) > 1 else "map50,room50").split(",")
    bad = 0
    for wl in names:
        worlds = [workloads.build_job(j)[0] for j in workloads.workload_jobs(wl, None)]
        for solver in ("product", "product_refined"):
            path = arbiter_run.cache_path(wl, solver)
            if not os.path.exists(path):
                print("%s %s: no cache (run scripts/arbiter_run.py --workload %s --solvers %s)" % (wl, solver, wl, solver))
                continue
            got = arbiter_run.pack(arbiter_run.run(solver, worlds, os.cpu_count() or 8))
            ref = np.load(path)
            same = all(np.array_equal(got[k], ref[k]) for k in ("solutions", "corridors", "sqp_iters", "admm_iters", "last_status"))
            print("%s %s: %s" % (wl, solver, "IDENTICAL" if same else "DIFFERENT"))
            bad += int(not same)
    sys.exit(1 if bad else 0)
