#!/bin/bash
# Diagnostic: does bench.py survive rocprofv3 with in-process set-up, and how long does a PMC pass take?
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
probe() {
  name=$1; shift
  timeout -s KILL 200 rocprofv3 "$@" > $R/gpurun_out/probe_$name.log 2>&1
  echo "== $name rc=$? t=$SECONDS : $(grep -c bad_variant $R/gpurun_out/probe_$name.log) bad_variant; files: $(ls -R $R/gpurun_out/probe_$name 2>/dev/null | grep -c csv); $(tail -1 $R/gpurun_out/probe_$name.log)"
}
probe inproc --kernel-trace --stats --output-format csv -d $R/gpurun_out/probe_inproc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --instances 2 --setup-procs 1
probe pmc --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/probe_pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --instances 4 --setup-procs 1
