#!/bin/bash
# The bench lines of all workloads (no tests, no profiles).  usage: gpu_bench_lines.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r04b}
O=gpurun_out/$TAG; mkdir -p $O
timeout 600 python bench.py > $O/bench_map100.json 2> $O/bench_map100.err; echo "bench map100 rc=$?"
for w in map50 synth1024 room50 agents100; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout 600 python bench.py --force-dist --no-cpu-baseline > $O/bench_map100_force_dist.json 2> $O/bench_map100_force_dist.err; echo "bench force-dist rc=$?"
timeout 300 python scripts/single_instance_times.py > $O/single_instance_times.txt 2>&1
