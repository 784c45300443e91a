"""A/B timing of kernel builds on one GPU box: bench.py's resident-input kernel time for each library given, interleaved
(A B A B ...) so that clock / thermal drift hits all alike.   python scripts/ab_bench.py [--workload map100] [--rounds 3] libA.so libB.so ..."""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--workload", default="map100")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=6)
args = ap.parse_args()
res = {l: [] for l in args.libs}
for r in range(args.rounds):
    for l in args.libs:
        env = dict(os.environ, CSDO_DIAG_LIB=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", args.workload, "--steps", str(args.steps),
                              "--warmup", "2", "--no-cpu-baseline", "--no-e2e", "--skip-single-instance"],
                             env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append(d["ms_per_step"])
        except Exception as e:
            print("ERR", l, out.stderr[-400:])
for l in args.libs:
    v = res[l]
    print("%-60s %s  min %.2f  mean %.2f" % (os.path.basename(l), " ".join("%.2f" % x for x in v), min(v) if v else -1, sum(v) / max(len(v), 1)))
