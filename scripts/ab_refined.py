"""A/B of the refined mode's kernel time (csdo_qp_parm::solve_refinement = 1) between library builds, interleaved on one box.
   python scripts/ab_refined.py [--workload map100] libA.so libB.so ..."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--workload", default="map100")
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--modes", default="1", help="solve_refinement values, comma separated (0 = off)")
args = ap.parse_args()
modes = [int(m) for m in args.modes.split(",")]
res = {(l, m): [] for l in args.libs for m in modes}
for r in range(args.rounds):
  for m in modes:
    for l in args.libs:
        env = dict(os.environ, CSDO_DIAG_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--solve-refinement", str(m), "--steps", "4",
                              "--warmup", "1", "--no-cpu-baseline", "--no-e2e", "--skip-single-instance"], env=env, capture_output=True, text=True)
        try:
            res[(l, m)].append(json.loads(out.stdout.strip().splitlines()[-1])["ms_per_step"])
        except Exception:
            print("ERR", l, out.stderr[-400:])
for (l, m), v in res.items():
    print("%-40s solve_refinement %d, %s: %s  min %.2f" % (os.path.basename(l), m, args.workload, " ".join("%.2f" % x for x in v), min(v) if v else -1))
