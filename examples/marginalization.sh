#!/bin/bash
# README.md "Example marginalization": burn in, sample every -f steps, print every node's most frequent block.  The reference
# parses -b / -f and drops them; --marginalize runs what its README describes, here with 256 independent chains pooled.
cd "$(dirname "$0")/.."
bipartitesbm-mcmc_amd/bin/mcmc -e tests/golden/bisbm-n_1000-ka_4-kb_6.edgelist -y 500 500 \
    -n 125 125 125 125 84 84 83 83 83 83 -z 4 6 -b 100000 -t 1000000 -f 10000 -E 1 --randomize -d 7 \
    --marginalize --rng philox --chains 256
