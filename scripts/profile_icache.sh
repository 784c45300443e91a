#!/bin/bash
# Instruction-cache counters of the agent kernel for several builds of the library.  usage (gpurun, repo root): profile_icache.sh <tag> lib1.so lib2.so ...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
O=$R/gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  export CSDO_DIAG_LIB=$R/$lib
  timeout -s KILL 600 rocprofv3 --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $O/ic_$n -o ic -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --setup-procs 1 --skip-single-instance > $O/ic_$n.log 2>&1
  python3 - $O/ic_$n $n <<'PY'
import csv, glob, os, sys
per = {}
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "dsqp_agent_kernel" in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
print(sys.argv[2], {k: "%.4g" % (sum(v) / 4.0) for k, v in sorted(per.items())})
PY
done
