#!/bin/bash
# chain parity reports + outlier fixtures (scripts/chain_parity.py) for both benchmark sets.  usage: gpu_chain.sh <tag>
# afterwards: cp gpurun_out/<tag>/chain_outliers_*.json tests/golden/ ; cp gpurun_out/<tag>/chain_*.json profiles/ (renamed per round)
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-chain}
O=gpurun_out/$TAG; mkdir -p $O
for w in map100 map50; do
  timeout 1200 python scripts/chain_parity.py --workload $w --out $O/chain_$w.json --fixture $O/chain_outliers_$w.json > $O/chain_$w.log 2>&1; echo "chain $w rc=$?"; tail -2 $O/chain_$w.log
done
