#!/bin/bash
# A/B of library builds whose results may differ in their last bits: every library is checked against the lane-serial build of the
# CURRENT source instead (first the last library given).   usage: gpu_ab_nobits.sh <tag> "<workloads>" lib...
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; WL=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
CSDO_DIAG_LIB=${@: -1} timeout 900 python scripts/gpu_regress.py --against-emu --workload map100,map50,room50 --instances 12 > $O/emu.txt 2>&1; grep "HIP vs lane-serial" $O/emu.txt
for w in $WL; do
  timeout 1200 python scripts/ab_bench.py --rounds 3 --workload $w "$@" > $O/ab_$w.txt 2>&1; echo "-- $w"; cat $O/ab_$w.txt
done
