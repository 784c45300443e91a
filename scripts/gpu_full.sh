#!/bin/bash
# Full GPU session: every -m gpu test, the bench lines of all workloads, rocprofv3 profiles.  usage: gpu_full.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r04}
O=gpurun_out/$TAG; mkdir -p $O
python - <<'PY' > $O/host_info.txt 2>&1
import os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
PY
cat $O/host_info.txt
if [ "${2:-}" != "nopytest" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -rA > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"
timeout 600 python bench.py > $O/bench_map100.json 2> $O/bench_map100.err; echo "bench map100 rc=$?"
for w in map50 synth1024 room50 agents100; do
  timeout 600 python bench.py --workload $w > $O/bench_$w.json 2> $O/bench_$w.err; echo "bench $w rc=$?"
done
timeout 600 python bench.py --force-dist --no-cpu-baseline > $O/bench_map100_force_dist.json 2> $O/bench_map100_force_dist.err; echo "bench force-dist rc=$?"
bash scripts/profile_round.sh $TAG map100 > $O/profile_map100.log 2>&1; tail -3 $O/profile_map100.log
bash scripts/profile_round.sh $TAG map50 > $O/profile_map50.log 2>&1; tail -3 $O/profile_map50.log
bash scripts/profile_round.sh $TAG synth1024 > $O/profile_synth1024.log 2>&1; tail -3 $O/profile_synth1024.log
bash scripts/profile_round.sh $TAG room50 > $O/profile_room50.log 2>&1; tail -3 $O/profile_room50.log
bash scripts/profile_round.sh $TAG agents100 > $O/profile_agents100.log 2>&1; tail -3 $O/profile_agents100.log
timeout 600 python scripts/profile_phases_sum.py 0,1,2,4,5,6,7,9 map100 > $O/phases_map100.txt 2>&1
timeout 600 python scripts/profile_phases_sum.py 0,2,3,4,5,6,7,9 map50 > $O/phases_map50.txt 2>&1
timeout 600 python scripts/stream_fractions.py map100 "0.08,0.27,0.65" > $O/stream_map100.txt 2>&1
timeout 900 python scripts/authors_sweep.py $O/authors_sweep.json > $O/authors_sweep.log 2>&1
timeout 300 python scripts/single_instance_times.py > $O/single_instance_times.txt 2>&1
timeout 600 python scripts/profile_phases_sum.py 1,2 room50 > $O/phases_room50_long.txt 2>&1
timeout 300 python scripts/group_times.py room50 > $O/group_times_room50.txt 2>&1
