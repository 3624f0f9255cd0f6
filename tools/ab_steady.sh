#!/bin/bash
# Diagnostic: like ab.sh, but every run first lets the chains do SPINUP sweeps (the regime a long run lives in).
#   SPINUP=150 VARIANTS="N0 N40" tools/ab_steady.sh
cd "$(dirname "$0")/.."
LIB=bipartitesbm-mcmc_amd/libbisbm_hip.so
cp $LIB /tmp/lib_keep.so
for v in ${VARIANTS:-A B}; do
  cp ab/$v.so $LIB
  python bench.py --chains ${CHAINS:-1024} --steps 3 --warmup 2 --spinup ${SPINUP:-150} --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v after ${SPINUP:-150} sweeps: %.4f us/step/chain  (%.3e updates/s, accepted %.3f)' % (d['roofline']['avg_launch_ms']*1e3/(d['roofline']['updates_per_launch']/${CHAINS:-1024}), d['value'], d['config']['accepted_fraction_last_timed_sweep']))"
done
cp /tmp/lib_keep.so $LIB
