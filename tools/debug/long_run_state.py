"""How the chains of the bench workload look after many sweeps (diagnostic): per-sweep kernel time and acceptance, and for
chain 0 the block sizes n_r, edge counts m_r and the log_q tier variable u = n_r / sqrt(m_r) of every block (the hot
step's closed form needs u > 24 for all four log_q arguments of a step).

    python tools/debug/long_run_state.py [sweeps] [chains]
"""
import importlib
import sys

import numpy as np

sys.path.insert(0, ".")
pkg = importlib.import_module("bipartitesbm-mcmc_amd")
syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 256
na = nb = 500000
E = 10_000_000
ka = kb = 32
n = na + nb
a, b = syn.planted_edges(na, nb, E, ka, kb, seed=1)
rowptr, col = pkg.edge_to_adj((a, b), n)
labels = syn.contiguous_labels(na, nb, ka, kb)
m = pkg.BlockModel(labels, syn.types_vector(na, nb), ka + kb, ka, kb, 1.0, (rowptr, col), n_chains=chains, rng="philox",
                   seed=20240229)
m.shuffle_bisbm()
mh = pkg.MetropolisHasting()
for s in range(sweeps):
    r = mh.anneal(m, pkg.constant_schedule, [1.0], n, 1 << 60)
    if s % 8 == 7 or s == sweeps - 1:
        mr, nr = np.array(m.get_m_r(0), dtype=np.float64), np.array(m.get_n_r(0), dtype=np.float64)
        u = nr / np.sqrt(np.maximum(mr, 1))
        print("sweep %3d  acceptance %.4f  kernel %.1f ms  u: min %.1f  blocks with u <= 24: %d of %d  (u <= 21: %d)  n_r min %d max %d"
              % (s, float(np.mean(r)), m.last_sweep_timing()[0], u.min(), int((u <= 24).sum()), len(u), int((u <= 21).sum()),
                 int(nr.min()), int(nr.max())), flush=True)
mr, nr = np.array(m.get_m_r(0)), np.array(m.get_n_r(0))
print("n_r", nr.tolist())
print("m_r", mr.tolist())
print("u  ", np.round(nr / np.sqrt(np.maximum(mr, 1)), 1).tolist())
