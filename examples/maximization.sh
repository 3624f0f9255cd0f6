#!/bin/bash
# README.md "Example maximization": anneal towards the most likely partition of southernWomen into 5 + 5 blocks.
# Default RNG mode = the reference's mt19937 streams: for -d 42 (and the hidden second engine seeded with 43) the label line is
# the reference's own.
cd "$(dirname "$0")/.."
bipartitesbm-mcmc_amd/bin/mcmc -e tests/golden/southernWomen.edgelist -n 4 4 3 4 3 3 3 3 3 2 -y 18 14 -z 5 5 \
    -t 32000 -x 100 -c exponential -a 10 0.1 -E 0.001 --randomize -d 42 --gen_seed 43
