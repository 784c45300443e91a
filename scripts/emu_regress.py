"""Bit-level regression of the device program through its lane-serial host build (tests/emu): solve a few worlds of both
benchmark sets, save everything the ABI returns, and later check that a changed program returns the same bits.
   python scripts/emu_regress.py --save /tmp/emu_ref.npz      python scripts/emu_regress.py --check /tmp/emu_ref.npz"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save", default=None)
    ap.add_argument("--check", default=None)
    ap.add_argument("--instances", type=int, default=4)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    from csdotrajectoryplanning_amd import workloads
    from tests import emu_lib
    worlds = []
    for name in ("map100", "map50"):
        worlds += [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs(name, args.instances), args.threads)]
    worlds += [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs("map100", 2, front="stand-in"), args.threads)]
    sols = emu_lib.solve_batch(worlds, 0, args.threads)
    out = {}
    for k, s in enumerate(sols):
        out["sol%d" % k], out["cor%d" % k] = s.solutions, s.corridors
        out["cnt%d" % k] = np.stack([s.sqp_iters, s.admm_iters, s.last_status])
    if args.save:
        np.savez(args.save, **out)
        print("saved", len(sols), "worlds,", int(sum(int(s.admm_iters.sum()) for s in sols)), "ADMM iterations")
    if args.check:
        ref = np.load(args.check)
        worst, bad = 0.0, 0
        for k in out:
            if not np.array_equal(out[k], ref[k]):
                bad += 1
                if out[k].dtype.kind == "f":
                    worst = max(worst, float(np.abs(out[k] - ref[k]).max()))
                else:
                    print("counts differ:", k, np.nonzero((out[k] != ref[k]).any(axis=0))[0][:10])
        print("IDENTICAL" if not bad else "DIFFERENT: %d arrays, max |d| %.3e" % (bad, worst))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
