#!/bin/bash
# Diagnostic only: rebuilds the library with pieces of the step removed (results are WRONG in those
# builds) and times a reduced bench, to attribute the per-step cost.  Restores the real build at the end.
#   ABLATE_SET="0 8"  CHAINS_SET="256 1024"  tools/ablate.sh
set -e
cd "$(dirname "$0")/.."
for abl in ${ABLATE_SET:-0 8}; do
  BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_ABLATE=$abl" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
  for chains in ${CHAINS_SET:-64}; do
    echo -n "ablate=$abl chains=$chains  "
    python bench.py --chains $chains --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('updates/s %.3e  us/step/chain %.3f' % (d['value'], $chains*1e6/d['value']))"
  done
done
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
