#!/bin/bash
# Diagnostic: like ab.sh on the config-5 shape (N = 4e6, E = 5e7, 64 + 64 blocks: the K > 32 variant of the production kernel).
#   VARIANTS="A B" ROUNDS=2 tools/ab_config5.sh
cd "$(dirname "$0")/.."
LIB=bipartitesbm-mcmc_amd/libbisbm_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${VARIANTS:-A B}; do
    cp ab/$v.so $LIB
    python bench.py --na 2000000 --nb 2000000 --edges 50000000 --ka 64 --kb 64 --chains ${CHAINS:-1024} --steps 3 --warmup 1 --spinup 1 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $r: %.4f us/step/chain  (%.3e updates/s)' % (d['roofline']['avg_launch_ms']*1e3/(d['roofline']['updates_per_launch']/${CHAINS:-1024}), d['value']))"
  done
done
cp /tmp/lib_keep.so $LIB
