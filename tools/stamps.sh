#!/bin/bash
# Diagnostic only: in-kernel time stamps (s_memtime, 10 ns ticks) per step stage and per chunk barrier.
#   CHAINS_SET="256 1024" tools/stamps.sh            (BISBM_SINGLE_STEPS=1 for the one-step-per-pass loop)
# A "step" in the output is one executed node update; with two steps per pass the stage times are per update too.
set -e
cd "$(dirname "$0")/.."
BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_STAMPS=${STAMPS_LEVEL:-1}" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
for chains in ${CHAINS_SET:-64}; do
  echo "== chains=$chains single=${BISBM_SINGLE_STEPS:-0}"
  python bench.py --chains $chains --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamps | tail -12
done
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
