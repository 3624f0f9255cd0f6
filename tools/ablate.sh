#!/bin/bash
# Diagnostic only: rebuilds the library with pieces of the step removed (results are WRONG in those
# builds) and times a reduced bench, to attribute the per-step cost.  Restores the real build at the end.
set -e
cd "$(dirname "$0")/.."
for abl in ${ABLATE_SET:-0 1 2 4 7}; do
  BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_ABLATE=$abl" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
  echo -n "ablate=$abl  "
  python bench.py --chains 64 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('updates/s %.3e  us/step/chain %.2f' % (d['value'], 64e6/d['value']))"
done
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
