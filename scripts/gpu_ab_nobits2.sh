#!/bin/bash
# A/B of library builds, timing only: usage gpu_ab_nobits2.sh <tag> "<workloads>" lib...  (+ the single-instance times of each library)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; WL=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
for w in $WL; do
  timeout 1200 python scripts/ab_bench.py --rounds 3 --workload $w "$@" > $O/ab_$w.txt 2>&1; echo "-- $w"; cat $O/ab_$w.txt
done
for lib in "$@"; do
  CSDO_DIAG_LIB=$lib timeout 300 python scripts/single_instance_times.py > $O/single_$(basename $lib).txt 2>&1; tail -n 1 $O/single_$(basename $lib).txt
  CSDO_DIAG_LIB=$lib timeout 300 python scripts/single_instance_times.py > $O/single2_$(basename $lib).txt 2>&1; tail -n 1 $O/single2_$(basename $lib).txt
done
