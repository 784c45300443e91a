#!/bin/bash
# Diagnostic: the phase-timer build (make prof) with one more define on the 512-thread / mode-0 kernel only:
#   build_prof_ablation.sh <tag> "<defines>"  ->  csdotrajectoryplanning_amd/libcsdo_hip_prof_<tag>.so  (CSDO_PROF_LIB selects it)
set -e
TAG=$1; EXTRA=${2:-}
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/csdotrajectoryplanning_amd/csrc; B=/tmp/csdo_prof_$TAG; mkdir -p $B
make -s -C $C prof
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-but-set-variable -Wno-unused-variable \
  -DCSDO_PROFILE_PHASES $EXTRA -DCSDO_V_BLOCK=512 -DCSDO_V_MODE=0 -DCSDO_V_SPLIT=1 -c $C/dsqp_variant.hip -o $B/variant_512_0_1.o
OBJS=$(ls $C/build/prof/*.o | grep -v variant_512_0_1)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/csdotrajectoryplanning_amd/libcsdo_hip_prof_$TAG.so $OBJS $B/variant_512_0_1.o \
  $C/build/bridge_host.o $C/build/front_end.o $C/build/aux_kernels.o
echo built libcsdo_hip_prof_$TAG.so
