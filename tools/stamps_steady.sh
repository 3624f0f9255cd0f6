#!/bin/bash
# Diagnostic: the per-stage stamps of tools/stamps.sh after SPINUP sweeps of the bench chains (default 150).
set -e
cd "$(dirname "$0")/.."
BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_STAMPS=1" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
python bench.py --chains ${CHAINS:-1024} --spinup ${SPINUP:-150} --steps 1 --warmup 0 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamps | tail -12
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
