"""Time of the agglomerative merge on the bench graph (BASELINE configs[3] shape: 512 chains, Ka = Kb = 32), diagnostic."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("bipartitesbm-mcmc_amd"); syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
na = nb = 500000; E = 10_000_000; ka = kb = 32; n = na + nb
a, b = syn.planted_edges(na, nb, E, ka, kb, seed=1); rowptr, col = pkg.edge_to_adj((a, b), n)
labels = syn.contiguous_labels(na, nb, ka, kb)
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 512
m = pkg.BlockModel(labels, syn.types_vector(na, nb), ka + kb, ka, kb, 1.0, (rowptr, col), n_chains=chains, rng="philox", seed=3)
m.shuffle_bisbm(); mh = pkg.MetropolisHasting()
mh.anneal(m, pkg.constant_schedule, [1.0], 2 * n, 1 << 60)
print("2 sweeps: %.1f ms kernel" % m.last_sweep_timing()[0], flush=True)
for da, db in [(4, 4), (8, 8)]:
    t = time.time(); m.agg_merge(da, db, 10); dt = time.time() - t
    print("agg_merge(-%d, -%d) of %d chains: %.2f s -> Ka, Kb = %d, %d" % (da, db, chains, dt, m.get_KA(), m.get_KB()), flush=True)
    t = time.time(); mh.anneal(m, pkg.constant_schedule, [1.0], n, 1 << 60)
    print("  one sweep after it: %.1f ms kernel, entropy[0] %.6e" % (m.last_sweep_timing()[0], m.entropy()[0]), flush=True)
