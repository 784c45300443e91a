#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-tail12}; mkdir -p $O
for w in map100 synth1024 agents100 room50; do
  timeout 1200 python scripts/ab_bench.py --rounds 2 --workload $w ab/lib_tail8.so ab/lib_tail12.so > $O/ab_$w.txt 2>&1; echo "-- $w"; cat $O/ab_$w.txt
done
CSDO_DIAG_LIB=ab/lib_tail8.so timeout 300 python scripts/single_instance_times.py > $O/single8.txt 2>&1; tail -n 1 $O/single8.txt
CSDO_DIAG_LIB=ab/lib_tail12.so timeout 300 python scripts/single_instance_times.py > $O/single12.txt 2>&1; tail -n 1 $O/single12.txt
