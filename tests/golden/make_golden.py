"""Generates tests/golden/*.npz from the oracle (there is no reference build to generate them from: SURVEY 8c).
Run from the repo root:  python tests/golden/make_golden.py
Each file holds the exact inputs of one small DO-phase problem and the oracle's outputs, including the
per-SQP-iteration trace, so later edits of the oracle (or another machine's libm) are caught by tests/test_oracle.py."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from csdotrajectoryplanning_amd import workloads  # noqa: E402
from tests import oracle_lib  # noqa: E402


def dump(name, world):
    sol = oracle_lib.solve(world, 1)
    meta, deltas, sols = oracle_lib.trace(world)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", name), x0_bar=world.x0_bar, plane_off=world.plane_off,
                        planes_t=world.planes["t"], planes_c=world.planes["c"], dimx=world.dimx, dimy=world.dimy,
                        obstacles=world.obstacles, solutions=sol.solutions, corridors=sol.corridors,
                        sqp_iters=sol.sqp_iters, admm_iters=sol.admm_iters, last_status=sol.last_status,
                        solver_status=sol.solver_status, initial_static_legal=sol.initial_static_legal,
                        trace_meta=meta, trace_delta=deltas, trace_sol=sols.astype(np.float64))
    print(name, world.Na, world.Nt, sol.sqp_iters, sol.last_status)


if __name__ == "__main__":
    w50, _ = workloads.build_world(workloads.MAP50_AGENTS25, seed=0, preprocess=oracle_lib.preprocess)
    dump("map50_agents0to5.npz", w50.subset(0, 6))
    dump("map50_agents15to17.npz", w50.subset(15, 18))
    w100, _ = workloads.build_world(workloads.MAP100_AGENTS50.format(0), seed=0, preprocess=oracle_lib.preprocess)
    dump("map100_agents0to3.npz", w100.subset(0, 4))
