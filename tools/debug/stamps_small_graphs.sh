#!/bin/bash
# Diagnostic: chunk-level stamps (time in a chunk's steps vs waiting for the feeder wave) on the reference's two data sets,
# where the stepping wave runs four / eight steps per pass -- is the feeder the limit there?  (n_1000: 375 cycles per step in
# the steps, 2.5 waiting: no.)
cd "$(dirname "$0")/../.."
BISBM_EXTRA_HIPCC_FLAGS="-DBISBM_STAMPS=2" python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
python3 tools/debug/small_graph_speed.py 2>&1 | grep -E "chunk tail|barrier wait|total|chains:"
python bipartitesbm-mcmc_amd/build.py --force > /dev/null 2>&1
