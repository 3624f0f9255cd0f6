#!/bin/bash
# Wall time of the re-hosted CLI on the n_1000 data set: the reference's default usage (one chain, mt19937-compat) and the
# engine's own mode (--rng philox, one chain and 256 chains).  2e7 steps = 20000 sweeps.
cd "$(dirname "$0")/../.."
EL=tests/golden/bisbm-n_1000-ka_4-kb_6.edgelist
ARGS="-e $EL -y 500 500 -n 125 125 125 125 84 84 83 83 83 83 -z 4 6 -t 20000000 -x 100000000 -c constant -a 1 -E 1 --randomize -d 7"
for mode in "--rng mt19937-compat" "--rng philox" "--rng philox --chains 256"; do
  s=$(date +%s%N); bipartitesbm-mcmc_amd/bin/mcmc $ARGS $mode > /dev/null 2> /tmp/cli_err.txt; e=$(date +%s%N)
  echo "$mode: $(( (e - s) / 1000000 )) ms wall; $(grep -i "acceptance" /tmp/cli_err.txt | head -1)"
done
