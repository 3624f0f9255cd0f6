#!/bin/bash
# Diagnostic: like ab.sh with every chain started on the planted partition (the regime of a long run near the mode) and the pass
# depth pinned (default: four steps per pass, the depth the library chooses there at 32 + 32 blocks).
#   VARIANTS="A B" ROUNDS=2 DEPTH=4 tools/ab_planted.sh
cd "$(dirname "$0")/.."
LIB=bipartitesbm-mcmc_amd/libbisbm_hip.so
cp $LIB /tmp/lib_keep.so
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${VARIANTS:-A B}; do
    cp ab/$v.so $LIB
    BISBM_PASS_DEPTH=${DEPTH:-4} python bench.py --planted-start --chains ${CHAINS:-1024} --steps 3 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v round $r: %.4f us/step/chain  (%.3e updates/s, accepted %.3f, steps per pass %s)' % (d['roofline']['avg_launch_ms']*1e3/(d['roofline']['updates_per_launch']/${CHAINS:-1024}), d['value'], d['config']['accepted_fraction_last_timed_sweep'], d['config'].get('per_launch_steps_per_pass')))"
  done
done
cp /tmp/lib_keep.so $LIB
