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
    ap.add_argument("--against-emu", action="store_true",
                    help="compare every array with the lane-serial host build of the same source (tests/emu), agent by agent")
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--solve-refinement", type=int, nargs="?", const=1, default=0, help="csdo_qp_parm::solve_refinement (1 or 2) on every world (the REFINE kernels)")
    args = ap.parse_args()
    import torch  # noqa: F401  (first: the HIP runtime of the process)
    from csdotrajectoryplanning_amd import workloads
    from csdotrajectoryplanning_amd.solver import DsqpHandle
    out = {}
    for name in args.workload.split(","):
        worlds = [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs(name, args.instances), 8)]
        if args.solve_refinement:
            worlds = [w.with_parm(solve_refinement=args.solve_refinement) for w in worlds]
        h = DsqpHandle(0)
        got = h.solve_batch(worlds)
        if args.against_emu:
            from tests import emu_lib
            ref = emu_lib.solve_batch(worlds, 0, args.threads)
            n_agents = n_diff = 0
            worst = 0.0
            for k, (g, r) in enumerate(zip(got, ref)):
                d = np.maximum(np.abs(g.solutions - r.solutions).reshape(g.solutions.shape[0], -1).max(axis=1),
                               np.abs(g.corridors - r.corridors).reshape(g.corridors.shape[0], -1).max(axis=1))
                cnt = (g.sqp_iters != r.sqp_iters) | (g.admm_iters != r.admm_iters) | (g.last_status != r.last_status)
                bits = np.array([not (np.array_equal(g.solutions[a], r.solutions[a]) and np.array_equal(g.corridors[a], r.corridors[a]))
                                 for a in range(g.solutions.shape[0])]) | cnt
                n_agents += len(d)
                n_diff += int(bits.sum())
                worst = max(worst, float(d.max()))
                if bits.any():
                    print("  world %d: agents with different bits %s (counts differ: %s), max |d| %.3g" % (
                        k, np.nonzero(bits)[0][:8].tolist(), np.nonzero(cnt)[0][:8].tolist(), float(d.max())))
            print("%s: HIP vs lane-serial build: %d of %d agents differ in some bit, max |d| = %.3g  -> %s" % (
                name, n_diff, n_agents, worst, "BIT-IDENTICAL" if n_diff == 0 else "DIFFERENT"))
        for k, s in enumerate(got):
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
