#!/bin/bash
# Build the whole library with extra device-compile flags into ab/lib_<tag>.so (A/B experiments: scripts/ab_bench.py).
#   usage: build_variant.sh <tag> "<extra hipcc flags>"
set -e
TAG=$1; EXTRA=${2:-}
R=$(cd "$(dirname "$0")/.." && pwd); C=$R/csdotrajectoryplanning_amd/csrc; B=/tmp/csdo_build_$TAG; mkdir -p $B $R/ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-but-set-variable -Wno-unused-variable $EXTRA"
pids=()
for v in 256_0_1 512_0_1 512_1_1 768_2_1 768_3_1 1024_3_1 256_0_3 512_0_3 512_1_3 768_2_3 768_3_3 1024_3_3 256_0_5 512_0_5 512_1_5 768_2_5 768_3_5 1024_3_5; do
  IFS=_ read b m s <<< "$v"
  /opt/rocm/bin/hipcc $FLAGS -DCSDO_V_BLOCK=$b -DCSDO_V_MODE=$m -DCSDO_V_SPLIT=$s -c $C/dsqp_variant.hip -o $B/variant_$v.o & pids+=($!)
done
/opt/rocm/bin/hipcc $FLAGS -c $C/dsqp_kernel.hip -o $B/dsqp_kernel.o & pids+=($!)
/opt/rocm/bin/hipcc $FLAGS -c $C/aux_kernels.hip -o $B/aux_kernels.o & pids+=($!)
/opt/rocm/bin/hipcc $FLAGS -c $C/capi.hip -o $B/capi.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/ab/lib_$TAG.so $B/*.o $C/build/bridge_host.o $C/build/front_end.o
echo built ab/lib_$TAG.so
