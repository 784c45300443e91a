#!/bin/bash
# quick GPU check: parity tests that pin the kernel + short bench lines.  usage: gpu_quick.sh <tag> [pytest -k expr]
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-q}; K=${2:-"not full_chain and not synthetic"}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q --tb=short -x -k "$K" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_map100.json 2> $O/bench_map100.err; echo "bench rc=$?"
timeout 300 python bench.py --workload map50 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_map50.json 2> $O/bench_map50.err; echo "bench50 rc=$?"
python - $O <<'PY'
import json,sys
for f in ("bench_map100.json","bench_map50.json"):
    try:
        d=json.load(open("%s/%s"%(sys.argv[1] if len(sys.argv)>1 else "gpurun_out/q",f)))
        print(f, "%.2f M it/s"%(d["value"]/1e6), "%.1f ms/step"%d["ms_per_step"], "single %.1f ms"%d["single_instance"]["do_phase_ms"]["solve_kernel"], [ (g["agents"],g["threads"],g["lds_residency_mode"],round(g["avg_ms"],1)) for g in d["config"]["launch_groups_rank0"]])
    except Exception as e: print(f, "ERR", e)
PY
