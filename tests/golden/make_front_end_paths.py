"""Coarse paths of this repository's own front end (csdo_front_end_plan) for the benchmark instances under
tests/golden/instances: the initial guesses of the measured workloads (SURVEY 8d "Initial guesses": the build's own host
front end once it exists).  One .npz per instance that the search solves; instances it does not solve are listed in
unsolved.json and keep the seeded stand-in generator (synth.py).

The search is deterministic for given limits (srand(seed) inside, no wall-clock decisions below the time limit); the time
limit here is generous so that it never decides.  tests/test_front_end.py re-plans some instances and compares.

usage: python tests/golden/make_front_end_paths.py [procs] [name filter (substring)] [--reference-rules OUT_DIR]
  --reference-rules OUT_DIR: csdo_front_end_parm::keep_off_lower_goals = 0 (the reference's rule set), results into OUT_DIR
  instead of tests/golden/front_end_paths (for the solved-count comparison in profiles/r03_front_end_rules.json)
"""
import json
import os
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from csdotrajectoryplanning_amd import config, front_end, instance, workloads  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "front_end_paths")
TIME_LIMIT_S = 120.0
KEEP_OFF = 1


def run(name):
    veh = config.vehicle_from_config()
    inst = instance.load_instance(os.path.join(workloads.INSTANCE_DIR, name), obs_radius=veh.obs_radius)
    parm = front_end.default_parm()
    parm.time_limit_s = TIME_LIMIT_S
    parm.keep_off_lower_goals = KEEP_OFF
    cp = front_end.plan(inst.starts, inst.goals, inst.dimx, inst.dimy, inst.obstacles, veh, parm)
    out = os.path.join(OUT, name.replace(".yaml", ".npz"))
    fallback = False
    if cp is None and KEEP_OFF:      # second attempt with the reference's rule set (csdo_front_end_parm::keep_off_lower_goals = 0)
        parm.keep_off_lower_goals = 0
        cp = front_end.plan(inst.starts, inst.goals, inst.dimx, inst.dimy, inst.obstacles, veh, parm)
        fallback = cp is not None
    if cp is None:
        if os.path.exists(out):
            os.remove(out)      # solved by an earlier version of the search
        return name, None
    np.savez_compressed(out, states=cp.states, actions=cp.actions,
                        path_off=cp.path_off, hl_expanded=cp.hl_expanded, ll_expanded=cp.ll_expanded)
    return name, (cp.seconds, cp.hl_expanded, cp.ll_expanded, fallback)


def _init(out, keep_off):
    global OUT, KEEP_OFF
    OUT, KEEP_OFF = out, keep_off


if __name__ == "__main__":
    args = [a for a in sys.argv[1:]]
    if "--reference-rules" in args:
        k = args.index("--reference-rules")
        OUT, KEEP_OFF = args[k + 1], 0
        del args[k:k + 2]
    procs = int(args[0]) if len(args) > 0 else 4
    pick = args[1] if len(args) > 1 else ""
    names = sorted(n for n in os.listdir(workloads.INSTANCE_DIR) if n.endswith(".yaml") and pick in n)
    os.makedirs(OUT, exist_ok=True)
    with ProcessPoolExecutor(procs, initializer=_init, initargs=(OUT, KEEP_OFF)) as ex:
        res = list(ex.map(run, names))
    unsolved = sorted(n for n, r in res if r is None)
    listed = {"time_limit_s": TIME_LIMIT_S, "unsolved": []}
    path = os.path.join(OUT, "unsolved.json")
    if pick and os.path.exists(path):          # a partial run keeps what it did not look at
        with open(path) as f:
            listed = json.load(f)
        listed["unsolved"] = [n for n in listed["unsolved"] if pick not in n]
    listed["unsolved"] = sorted(set(listed["unsolved"]) | set(unsolved))
    fb = set(n for n in listed.get("planned_with_reference_rules", []) if not (pick in n)) | set(n for n, r in res if r is not None and r[3])
    listed["planned_with_reference_rules"] = sorted(fb)
    with open(path, "w") as f:
        json.dump(listed, f, indent=1)
    ok = [r for _, r in res if r is not None] or [(0.0, 0, 0, False)]
    print("solved %d of %d; search seconds mean %.2f max %.2f" % (len(ok), len(res), np.mean([r[0] for r in ok]),
                                                                   max(r[0] for r in ok)))
    print("unsolved:", unsolved)
