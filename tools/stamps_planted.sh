#!/bin/bash
# Diagnostic: chunk-level stamps (time in the steps vs time waiting for the feeder wave) with every chain on the planted
# partition -- the regime where most steps end at the r == s test and the feeder has least time per chunk.
set -e
cd "$(dirname "$0")/.."
BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_STAMPS=${STAMPS_LEVEL:-2}" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
python bench.py --chains ${CHAINS:-1024} --planted-start --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamps | tail -12
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
