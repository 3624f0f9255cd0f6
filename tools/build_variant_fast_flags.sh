#!/bin/bash
# Diagnostic: build ab/NAME.so with extra hipcc flags for the production-kernel TU only (the other TUs as usual).
#   tools/build_variant_fast_flags.sh NAME -mllvm -some-flag
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p ab /tmp/bv_$NAME
C=bipartitesbm-mcmc_amd/csrc
COMMON="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -pthread -Wno-unused-function -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -amdgpu-atomic-optimizer-strategy=None"
for f in bisbm_kernels.hip bisbm_runtime.hip bisbm_io.cpp; do
  [ -f /tmp/bv_common_${f%.*}.o ] && [ /tmp/bv_common_${f%.*}.o -nt $C/$f ] || /opt/rocm/bin/hipcc $COMMON -c $C/$f -o /tmp/bv_common_${f%.*}.o
done
/opt/rocm/bin/hipcc $COMMON "$@" -c $C/bisbm_sweep_fast.hip -o /tmp/bv_$NAME/fast.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ab/$NAME.so /tmp/bv_common_bisbm_kernels.o /tmp/bv_common_bisbm_runtime.o /tmp/bv_common_bisbm_io.o /tmp/bv_$NAME/fast.o
echo ab/$NAME.so
