"""Per-QP parity of a backend (gpu | emu) against the oracle: every agent of a benchmark set solved with QpParm.max_iter
= 1 (one QP from x0_bar, identical boxes by construction) and = 2 (one corridor refresh in between).  Prints one JSON
object; --out writes it under profiles/.   python scripts/first_qp_parity.py --backend gpu --workload map100"""
import argparse
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def with_max_iter(world, k):
    from csdotrajectoryplanning_amd.problem import World
    from csdotrajectoryplanning_amd.abi import QpParm
    p = QpParm.from_buffer_copy(bytes(world.parm))
    p.max_iter = float(k)
    return World(world.x0_bar, world.plane_off, world.planes, world.dimx, world.dimy, world.obstacles, world.veh, p)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=("gpu", "emu"), default="gpu")
    ap.add_argument("--workload", choices=("map100", "map50", "synth1024", "room50", "agents100"), default="map100")
    ap.add_argument("--instances", type=int, default=None)
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 8)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from csdotrajectoryplanning_amd import workloads
    from tests import oracle_lib
    worlds = [w for w, _ in workloads.build_jobs_parallel(workloads.workload_jobs(args.workload, args.instances),
                                                          min(args.threads, 32))]
    if args.backend == "gpu":
        from csdotrajectoryplanning_amd.solver import DsqpHandle
        h = DsqpHandle(0)
        solve = h.solve_batch
    else:
        from tests import emu_lib
        solve = lambda ws: emu_lib.solve_batch(ws, 0, args.threads)
    report = {"backend": args.backend, "workload": args.workload, "agents": int(sum(w.Na for w in worlds))}
    for k in (1, 2, 10):
        ws = [with_max_iter(w, k) for w in worlds]
        got = solve(ws)
        ref = oracle_lib.solve_batch(ws, args.threads)
        d, dc, same, rows = [], [], [], []
        for wi, (g, r) in enumerate(zip(got, ref)):
            dd = np.abs(g.solutions - r.solutions).max(axis=(1, 2))
            cc = np.abs(g.corridors - r.corridors).max(axis=(1, 2))
            ss = (g.sqp_iters == r.sqp_iters) & (g.admm_iters == r.admm_iters) & (g.last_status == r.last_status)
            d.append(dd); dc.append(cc); same.append(ss)
            for a in np.nonzero(~ss | (dd > 1e-6))[0]:
                rows.append({"world": wi, "agent": int(a), "d": float(dd[a]), "d_corridor": float(cc[a]),
                             "admm": [int(g.admm_iters[a]), int(r.admm_iters[a])],
                             "sqp": [int(g.sqp_iters[a]), int(r.sqp_iters[a])],
                             "status": [int(g.last_status[a]), int(r.last_status[a])]})
        d, dc, same = np.concatenate(d), np.concatenate(dc), np.concatenate(same)
        report["max_iter_%d" % k] = {
            "same_counts": int(same.sum()), "max_d": float(d.max()), "median_d": float(np.median(d)),
            "n_gt_1e-6": int((d > 1e-6).sum()), "n_gt_1e-5": int((d > 1e-5).sum()), "n_gt_1e-4": int((d > 1e-4).sum()),
            "n_gt_1e-3": int((d > 1e-3).sum()), "n_gt_2e-2": int((d > 2e-2).sum()),
            "n_box_step_flipped": int((dc > 0.05).sum()),          # a box edge moved by a 0.1 m growth step
            "n_gt_1e-4_without_box_flip": int(((d > 1e-4) & (dc < 0.05)).sum()),
            "listed": sorted(rows, key=lambda r: -r["d"])[:40]}
    s = json.dumps(report, indent=1)
    print(s)
    if args.out:
        with open(args.out, "w") as f:
            f.write(s + "\n")


if __name__ == "__main__":
    main()
