#!/bin/bash
# Diagnostic: alternate two prebuilt libraries (ab/A.so, ab/B.so) on the same box and print the sweep-kernel time.
#   CHAINS=1024 ROUNDS=2 tools/ab.sh
cd "$(dirname "$0")/.."
LIB=bipartitesbm-mcmc_amd/libbisbm_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${VARIANTS:-A B}; do
    cp ab/$v.so $LIB
    python bench.py --chains ${CHAINS:-1024} --steps 3 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $r: %.4f us/step/chain  (%.3e updates/s)' % (d['roofline']['avg_launch_ms']*1e3/(d['roofline']['updates_per_launch']/${CHAINS:-1024}), d['value']))"
  done
done
cp /tmp/lib_keep.so $LIB
