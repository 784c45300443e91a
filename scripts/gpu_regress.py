"""Bit-level regression of the GPU library: solve a benchmark set with the library CSDO_DIAG_LIB names (default: the shipped
one), save everything the ABI returns, and check another build against it on the same box.
   python scripts/gpu_regress.py --save ab/gpu_ref.json [--workload map100]     CSDO_DIAG_LIB=ab/lib_X.so python scripts/gpu_regress.py --check ab/gpu_ref.json"""
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
    ap.add_argument("--workload", default="map100,map50")
    ap.add_argument("--instances", type=int, default=None)
    args = ap.parse_args()
    import torch  # noqa: F401  (first: the HIP runtime of the process)
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    out = {}
    for name in args.workload.split(","):
        worlds = [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs(name, args.instances), 8)]
        h = DsqpHandle(0)
        for k, s in enumerate(h.solve_batch(worlds)):
            out["%s_sol%d" % (name, k)], out["%s_cor%d" % (name, k)] = s.solutions, s.corridors
            out["%s_cnt%d" % (name, k)] = np.stack([s.sqp_iters, s.admm_iters, s.last_status])
    import hashlib
    import json
    digest = {k: hashlib.sha256(np.ascontiguousarray(v).tobytes()).hexdigest() for k, v in out.items()}
    if args.save:
        with open(args.save, "w") as f:
            json.dump(digest, f)
        print("saved the digests of", len(out) // 3, "worlds")
    if args.check:
        with open(args.check) as f:
            ref = json.load(f)
        bad = sorted(k for k in digest if digest[k] != ref.get(k))
        print("IDENTICAL" if not bad else "DIFFERENT: %d of %d arrays, e.g. %s" % (len(bad), len(digest), bad[:4]))
        sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
