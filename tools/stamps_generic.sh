#!/bin/bash
# Diagnostic only: in-kernel time stamps (s_memtime, 10 ns ticks) per stage of the GENERIC kernel's step (mt19937-compat
# mode, or Philox with more than 64 blocks of a type), printed per launch by the library on stderr.
#   tools/stamps_generic.sh
set -e
cd "$(dirname "$0")/.."
BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_GSTAMPS=1" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
python3 tools/debug/compat_speed.py 0.3 2>&1 | grep -v "^$"
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
