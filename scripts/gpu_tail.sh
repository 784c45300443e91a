#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-tail}; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_sets.py -x -q -k "lane_serial and (room50 or map50)" > $O/pytest.log 2>&1; tail -n 5 $O/pytest.log
for w in room50 map100 map50; do
  timeout 1200 python scripts/ab_bench.py --rounds 2 --workload $w ab/lib_packed.so ab/lib_tail8.so > $O/ab_$w.txt 2>&1; echo "-- $w"; cat $O/ab_$w.txt
done
