#!/bin/bash
# round 5, first GPU call: is the HIP build bit-identical to the lane-serial build with the shared trigonometry, and what do
# the two forms of its call sites (inlined / real calls) cost against the round-4 library
set -u
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r5_first; mkdir -p $O
for lib in ab/lib_xm_call.so ab/lib_xm_inline.so; do
  CSDO_DIAG_LIB=$lib timeout 900 python scripts/gpu_regress.py --against-emu --workload map100,map50,room50,agents100,synth1024 > $O/emu_$(basename $lib .so).txt 2>&1
  grep "HIP vs lane-serial" $O/emu_$(basename $lib .so).txt
done
timeout 900 python scripts/ab_bench.py --rounds 3 ab/lib_r4.so ab/lib_xm_inline.so ab/lib_xm_call.so > $O/ab_map100.txt 2>&1; cat $O/ab_map100.txt
timeout 600 python scripts/ab_bench.py --rounds 2 --workload room50 ab/lib_r4.so ab/lib_xm_inline.so ab/lib_xm_call.so > $O/ab_room50.txt 2>&1; cat $O/ab_room50.txt
timeout 600 python scripts/ab_bench.py --rounds 2 --workload map50 ab/lib_r4.so ab/lib_xm_inline.so ab/lib_xm_call.so > $O/ab_map50.txt 2>&1; cat $O/ab_map50.txt
