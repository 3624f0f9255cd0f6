#!/bin/bash
# --merge (mcmc_main.cc:349-399): one block per node, staged merges with greedy sweeps in between down to -z 4 6, then the final
# anneal.  1000 blocks at the start: the library runs its wide mode until 256 blocks are left.
cd "$(dirname "$0")/.."
bipartitesbm-mcmc_amd/bin/mcmc -e tests/golden/bisbm-n_1000-ka_4-kb_6.edgelist -y 500 500 -n 500 500 -z 4 6 --merge \
    -t 10000 -x 100000 -c abrupt_cool -a 100 -E 1 -d 42 --gen_seed 43
