#!/bin/bash
# Diagnostic: build a variant of the library into ab/NAME.so without touching the product library (same units and flags as the
# product build, compiled side by side: bipartitesbm-mcmc_amd/build.py --variant).
#   tools/build_variant.sh NAME [extra hipcc flags...]      e.g.  tools/build_variant.sh stamps1 -DBISBM_STAMPS=1
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p ab
python3 bipartitesbm-mcmc_amd/build.py --variant ab/$NAME.so "$@"
