#!/bin/bash
# Diagnostic only: in-kernel time stamps (s_memtime, 10 ns ticks) per step stage and per chunk barrier.
#   CHAINS_SET="256 1024" tools/stamps.sh
set -e
cd "$(dirname "$0")/.."
BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_STAMPS=${STAMPS_LEVEL:-1}" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
for chains in ${CHAINS_SET:-64}; do
  echo "== chains=$chains"
  python bench.py --chains $chains --steps 1 --warmup 1 --no-cpu-baseline 2>&1 >/dev/null | grep stamps | tail -12
done
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
