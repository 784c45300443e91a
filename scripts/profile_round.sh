#!/bin/bash
# Profiling call of a round (run through gpurun from the repo root): kernel trace + separate PMC passes.
#   usage: profile_round.sh <tag> [workload] [solve_refinement]
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r02}
WL=${2:-map100}
cd /tmp && export TMPDIR=/tmp
REFINE=${3:-0}     # csdo_qp_parm::solve_refinement of the profiled run (the REFINE kernels): summaries are labelled <workload>_refine<m>
ARGS="--workload $WL --steps 3 --warmup 1 --no-cpu-baseline --no-e2e --setup-procs 1 --skip-single-instance"
O=$R/gpurun_out/${TAG}_${WL}
if [ "$REFINE" != "0" ]; then ARGS="$ARGS --solve-refinement $REFINE"; O=${O}_refine$REFINE; fi
mkdir -p $O
timeout -s KILL 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o trace -- python3 $R/bench.py $ARGS > $O/trace.log 2>&1
timeout -s KILL 900 rocprofv3 --output-format csv --pmc FETCH_SIZE -d $O/fetch -o fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
timeout -s KILL 900 rocprofv3 --output-format csv --pmc WRITE_SIZE -d $O/write -o write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
timeout -s KILL 900 rocprofv3 --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $O/sq -o sq -- python3 $R/bench.py $ARGS > $O/sq.log 2>&1
# L2 (per XCD) hit rate: TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) (MI355X_MICROARCH.md, L2 section); requests leaving the L2 for the fabric
timeout -s KILL 900 rocprofv3 --output-format csv --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $O/tcc -o tcc -- python3 $R/bench.py $ARGS > $O/tcc.log 2>&1
timeout -s KILL 900 rocprofv3 --output-format csv --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum -d $O/tcc2 -o tcc2 -- python3 $R/bench.py $ARGS > $O/tcc2.log 2>&1
# the per-dispatch trace of a 3000-workgroup persistent launch is small, but drop anything big
find $O -name "*.csv" -size +4M -delete
tail -n 1 $O/trace.log | cut -c1-300
ls -R $O | head -30
