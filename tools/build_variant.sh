#!/bin/bash
# Diagnostic: build a variant of the library into ab/NAME.so without touching the product library.
#   tools/build_variant.sh NAME [extra hipcc flags...]      e.g.  tools/build_variant.sh stamps1 -DBISBM_STAMPS=1
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p ab
C=bipartitesbm-mcmc_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -pthread -Wall -Wno-unused-function \
  -mllvm -structurizecfg-skip-uniform-regions=true -mllvm -amdgpu-atomic-optimizer-strategy=None "$@" \
  -o ab/$NAME.so $C/bisbm_kernels.hip $C/bisbm_sweep_fast.hip $C/bisbm_runtime.hip $C/bisbm_io.cpp
echo ab/$NAME.so
