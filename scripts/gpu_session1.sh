#!/bin/bash
# GPU session: tests, per-QP parity reports, the three workloads' bench lines, a kernel trace of the map50 workload.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/s1
export TMPDIR=/tmp
O=gpurun_out/s1
timeout 1500 python -m pytest tests -m gpu -q --tb=short -rA > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
tail -40 $O/pytest_gpu.log
timeout 600 python scripts/first_qp_parity.py --backend gpu --workload map100 --out $O/first_qp_map100.json > /dev/null 2> $O/first_qp_map100.err
timeout 600 python scripts/first_qp_parity.py --backend gpu --workload map50 --out $O/first_qp_map50.json > /dev/null 2> $O/first_qp_map50.err
timeout 600 python bench.py > $O/bench_map100.json 2> $O/bench_map100.err; echo "bench map100 rc=$?"
timeout 600 python bench.py --workload map50 > $O/bench_map50.json 2> $O/bench_map50.err; echo "bench map50 rc=$?"
timeout 600 python bench.py --workload synth1024 > $O/bench_synth1024.json 2> $O/bench_synth1024.err; echo "bench synth rc=$?"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OLDPWD/$O/prof_map50 -o trace -- python3 $OLDPWD/bench.py --workload map50 --steps 5 --warmup 1 --setup-procs 1 --skip-single-instance --no-cpu-baseline --no-e2e > $OLDPWD/$O/prof_map50.log 2>&1); echo "rocprof rc=$?"
find $O/prof_map50 -name "*.db" -delete 2>/dev/null
find $O/prof_map50 -name "*kernel_trace.csv" -size +8M -delete 2>/dev/null
ls -la $O $O/prof_map50 2>/dev/null | head -40
