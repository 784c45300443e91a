"""The authors' own benchmark sweep (scripts/test_through_benchmark.sh:19-22 of the reference: map100by100 x {25, 30, 35, 40, 50}
agents x {obstacle, empty}, 60 instances each, PBS time limit 20 s) as data fixtures: the instance files (input data of the
reference's benchmark set, copied unchanged) under tests/golden/instances_sweep/ and the coarse paths this repository's front end
finds for them within the authors' 20 s under tests/golden/front_end_paths_sweep/ (default rule set first, the reference's rule
set for what that does not solve; unsolved.json lists what neither solves).  The agents50 / obstacle family is the map100 workload
and stays where it is (tests/golden/instances, front_end_paths).

usage (build container, /root/reference present): python tests/golden/make_sweep_fixtures.py [procs]"""
import json
import os
import shutil
import sys
from concurrent.futures import ProcessPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from csdotrajectoryplanning_amd import config, front_end, instance  # noqa: E402

REF = "/root/reference/benchmark/map100by100"
INST = os.path.join(ROOT, "tests", "golden", "instances_sweep")
OUT = os.path.join(ROOT, "tests", "golden", "front_end_paths_sweep")
FAMILIES = [(n, k) for n in (25, 30, 35, 40, 50) for k in ("obstacle", "empty") if not (n == 50 and k == "obstacle")]
TIME_LIMIT_S = 20.0      # csdo.cc:100 / test_through_benchmark.sh:17


def run(name):
    veh = config.vehicle_from_config()
    inst = instance.load_instance(os.path.join(INST, name), obs_radius=veh.obs_radius)
    parm = front_end.default_parm()
    parm.time_limit_s = TIME_LIMIT_S
    parm.keep_off_lower_goals = 1
    cp = front_end.plan(inst.starts, inst.goals, inst.dimx, inst.dimy, inst.obstacles, veh, parm)
    rules = "default"
    if cp is None:
        parm.keep_off_lower_goals = 0
        cp = front_end.plan(inst.starts, inst.goals, inst.dimx, inst.dimy, inst.obstacles, veh, parm)
        rules = "reference"
    if cp is None:
        return name, None
    np.savez_compressed(os.path.join(OUT, name.replace(".yaml", ".npz")), states=cp.states, actions=cp.actions,
                        path_off=cp.path_off, hl_expanded=cp.hl_expanded, ll_expanded=cp.ll_expanded, seconds=cp.seconds)
    return name, (float(cp.seconds), int(cp.hl_expanded), int(cp.ll_expanded), rules)


if __name__ == "__main__":
    procs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    os.makedirs(INST, exist_ok=True)
    os.makedirs(OUT, exist_ok=True)
    names = []
    for n, kind in FAMILIES:
        src = os.path.join(REF, "agents%d" % n, kind)
        for f in sorted(os.listdir(src)):
            if f.endswith(".yaml"):
                shutil.copyfile(os.path.join(src, f), os.path.join(INST, f))
                names.append(f)
    with ProcessPoolExecutor(procs) as ex:
        res = list(ex.map(run, names, chunksize=4))
    table = {}
    for name, r in res:
        fam = name.rsplit("_ex", 1)[0]
        t = table.setdefault(fam, {"instances": 0, "solved": 0, "with_reference_rules": 0, "search_seconds": []})
        t["instances"] += 1
        if r is not None:
            t["solved"] += 1
            t["with_reference_rules"] += int(r[3] == "reference")
            t["search_seconds"].append(r[0])
    for t in table.values():
        s = t.pop("search_seconds")
        t["search_seconds_mean"] = float(np.mean(s)) if s else None
        t["search_seconds_max"] = float(np.max(s)) if s else None
    with open(os.path.join(OUT, "unsolved.json"), "w") as f:
        json.dump({"time_limit_s": TIME_LIMIT_S, "unsolved": sorted(n for n, r in res if r is None), "families": table}, f, indent=1)
    print(json.dumps(table, indent=1))
