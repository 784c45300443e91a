#!/bin/bash
# round 5: rocprofv3 kernel trace + PMC passes (HBM bytes, SQ, L2 hit rate) for all five workloads, bench lines, phase profile
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r05}
O=gpurun_out/$TAG; mkdir -p $O
for w in map100 map50 synth1024 room50 agents100; do
  bash scripts/profile_round.sh $TAG $w > $O/profile_$w.log 2>&1; tail -n 2 $O/profile_$w.log | cut -c1-200
done
for w in map100 map50 synth1024 room50 agents100; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
