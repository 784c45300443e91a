#!/bin/bash
# Profiling call of a round (run through gpurun from the repo root): kernel trace + separate PMC passes.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --no-cpu-baseline --setup-procs 1 --skip-single-instance"
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o trace -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_trace.log 2>&1
timeout -s KILL 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_fetch -o fetch -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_fetch.log 2>&1
timeout -s KILL 900 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_write -o write -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_write.log 2>&1
timeout -s KILL 900 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $R/gpurun_out/${TAG}_sq -o sq -- python3 $R/bench.py $ARGS > $R/gpurun_out/${TAG}_sq.log 2>&1
tail -n 2 $R/gpurun_out/${TAG}_trace.log | cut -c1-400
ls -R $R/gpurun_out/${TAG}_trace | head -20
