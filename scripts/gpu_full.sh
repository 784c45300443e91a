#!/bin/bash
# Full GPU session: every -m gpu test, the three workloads' bench lines, parity reports, rocprofv3 profiles.  usage: gpu_full.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-r02}
O=gpurun_out/$TAG; mkdir -p $O
python - <<'PY' > $O/host_info.txt 2>&1
import os
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
PY
cat $O/host_info.txt
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 600 python scripts/first_qp_parity.py --backend gpu --workload map100 --out $O/parity_map100.json > /dev/null 2> $O/parity_map100.err
timeout 600 python scripts/first_qp_parity.py --backend gpu --workload map50 --out $O/parity_map50.json > /dev/null 2> $O/parity_map50.err
timeout 600 python bench.py > $O/bench_map100.json 2> $O/bench_map100.err; echo "bench map100 rc=$?"
timeout 600 python bench.py --workload map50 > $O/bench_map50.json 2> $O/bench_map50.err; echo "bench map50 rc=$?"
timeout 600 python bench.py --workload synth1024 > $O/bench_synth1024.json 2> $O/bench_synth1024.err; echo "bench synth rc=$?"
bash scripts/profile_round.sh $TAG map100 > $O/profile_map100.log 2>&1; tail -3 $O/profile_map100.log
bash scripts/profile_round.sh $TAG map50 > $O/profile_map50.log 2>&1; tail -3 $O/profile_map50.log
