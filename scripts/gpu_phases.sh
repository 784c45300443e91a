#!/bin/bash
# Diagnostic GPU call: quick parity + bench lines, phase profile (prof build) and per-agent times.  usage: gpu_phases.sh <tag>
set -u
cd "${GRAFT_REPO_ROOT:-.}"
TAG=${1:-ph}
O=gpurun_out/$TAG; mkdir -p $O
bash scripts/gpu_quick.sh $TAG "first_qp or aux or pipeline or residency"
timeout 600 python scripts/profile_phases_sum.py 0,1,2,4,5,6,7,9 map100 > $O/phases_map100.txt 2>&1; tail -30 $O/phases_map100.txt
timeout 600 python scripts/profile_phases_sum.py 0,2,3,4,5,6,7,9 map50 > $O/phases_map50.txt 2>&1
timeout 600 python scripts/agent_times.py map100 $O/agent_times_map100.npz > $O/agent_times.log 2>&1
timeout 600 python scripts/agent_times.py map50 $O/agent_times_map50.npz >> $O/agent_times.log 2>&1
cat $O/agent_times.log
