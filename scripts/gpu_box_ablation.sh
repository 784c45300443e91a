#!/bin/bash
# Diagnostic: what the pieces of a safe box cost - phase-timer builds that do one piece twice (same results), map100 instances
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/${1:-box_abl}; mkdir -p $O
for t in "" _box2x_ALL _box2x_PASSES _box2x_REPLAY _box2x_CULL; do
  CSDO_PROF_LIB=libcsdo_hip_prof$t.so timeout 600 python scripts/profile_phases_sum.py 0,1,2,4,5,6,7,9 map100 > $O/phases$t.txt 2>&1
  echo "prof$t: $(sed -n 1p $O/phases$t.txt | cut -c1-80)"; grep -m1 "cycles per SQP iteration: corridor" $O/phases$t.txt
done
