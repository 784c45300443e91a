#!/bin/bash
# round 5: the whole GPU suite, smoke, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-r5_suite}; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --tb=short -rA ${PYTEST_K:+-k "$PYTEST_K"} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -6 $O/pytest_gpu.log; grep -E "^(FAILED|ERROR)" $O/pytest_gpu.log | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
timeout 600 python bench.py > $O/bench_map100.json 2> $O/bench_map100.err; echo "bench map100 rc=$?"
python - $O <<'PY'
import json,sys
d=json.load(open(sys.argv[1]+"/bench_map100.json"))
print("map100 %.2f M it/s %.2f ms/step; e2e %.1f ms (kernels %.1f); single %.2f ms" % (d["value"]/1e6, d["ms_per_step"], d["do_phase_e2e"]["total_ms"], d["do_phase_e2e"]["solve_kernels_ms"], d["single_instance"]["do_phase_ms"]["solve_kernel"]))
PY
