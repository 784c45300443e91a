#!/bin/bash
# chain parity reports + outlier fixtures for the two regime workloads.  usage: gpu_chain_regimes.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-chainr}
O=gpurun_out/$TAG; mkdir -p $O
for w in room50 agents100; do
  timeout 1500 python scripts/chain_parity.py --workload $w --out $O/chain_$w.json --fixture $O/chain_outliers_$w.json > $O/chain_$w.log 2>&1; echo "chain $w rc=$?"; tail -2 $O/chain_$w.log | cut -c1-400
done
