#!/bin/bash
# A/B of library builds on one box: usage gpu_ab.sh <tag> "<workloads>" lib...   (bits checked against the first library)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=$1; WL=$2; shift 2
O=gpurun_out/$TAG; mkdir -p $O
CSDO_DIAG_LIB=$1 timeout 600 python scripts/gpu_regress.py --save $O/ref.json --workload map100,map50,room50 > $O/regress.txt 2>&1
for lib in "${@:2}"; do
  echo "== $lib" >> $O/regress.txt
  CSDO_DIAG_LIB=$lib timeout 600 python scripts/gpu_regress.py --check $O/ref.json --workload map100,map50,room50 >> $O/regress.txt 2>&1
done
grep "IDENTICAL\|DIFFERENT\|==" $O/regress.txt
for w in $WL; do
  timeout 1200 python scripts/ab_bench.py --rounds 3 --workload $w "$@" > $O/ab_$w.txt 2>&1; echo "-- $w"; cat $O/ab_$w.txt
done
