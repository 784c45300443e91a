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
timeout 600 python scripts/profile_phases_sum.py 0,1,2,4,5,6,7,9 map100 > $O/phases_map100.txt 2>&1
timeout 600 python scripts/profile_phases_sum.py 0,2,3,4,5,6,7,9 map50 > $O/phases_map50.txt 2>&1
timeout 600 python scripts/profile_phases_sum.py 1,2 room50 > $O/phases_room50_long.txt 2>&1
timeout 300 python scripts/single_instance_times.py > $O/single_instance_times.txt 2>&1
timeout 300 python scripts/group_times.py room50 > $O/group_times_room50.txt 2>&1
timeout 600 python bench.py --force-dist --no-cpu-baseline > $O/bench_map100_force_dist.json 2> $O/bench_map100_force_dist.err; echo "bench force-dist rc=$?"
