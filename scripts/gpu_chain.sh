#!/bin/bash
# chain parity reports (scripts/chain_parity.py) for both benchmark sets + a short bench line.  usage: gpu_chain.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-chain}
O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python scripts/chain_parity.py --workload map100 --out $O/chain_map100.json > $O/chain_map100.log 2>&1; echo "chain map100 rc=$?"; tail -3 $O/chain_map100.log
timeout 900 python scripts/chain_parity.py --workload map50 --out $O/chain_map50.json > $O/chain_map50.log 2>&1; echo "chain map50 rc=$?"; tail -3 $O/chain_map50.log
timeout 300 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_map100.json 2> $O/bench_map100.err; echo "bench rc=$?"
tail -c 600 $O/bench_map100.json
