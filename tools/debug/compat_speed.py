"""Per-step time of the mt19937-compat path (the re-hosted CLI's default: one chain, the reference's RNG streams) on the
reference's own data sets, next to the Philox path on one chain.  Diagnostic.  argv[1] = sweeps scale (default 1)."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("bipartitesbm-mcmc_amd"); syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
G = os.path.join("tests", "golden")
for name, path, na, nb, ka, kb, eps, sweeps in (("southernWomen", "southernWomen.edgelist", 18, 14, 5, 5, 0.001, 20000),
                                                ("n_1000", "bisbm-n_1000-ka_4-kb_6.edgelist", 500, 500, 4, 6, 1.0, 1000)):
    n = na + nb
    sweeps = max(1, int(sweeps * scale))
    rowptr, col = pkg.load_graph(os.path.join(G, path), n)
    lab = syn.contiguous_labels(na, nb, ka, kb)
    for rng, chains in (("mt19937-compat", 1), ("mt19937-compat", 64), ("philox", 1)):
        m = pkg.BlockModel(lab, syn.types_vector(na, nb), ka + kb, ka, kb, eps, (rowptr, col), n_chains=chains, rng=rng, seed=1)
        m.shuffle_bisbm(); mh = pkg.MetropolisHasting()
        mh.anneal(m, pkg.constant_schedule, [1.0], max(1, sweeps // 10) * n, 1 << 60)
        r = mh.anneal(m, pkg.constant_schedule, [1.0], sweeps * n, 1 << 60)
        ms, upd = m.last_sweep_timing()
        print("%-14s %-15s %3d chains: %8.1f ms for %d sweeps -> %.3f us per step per chain, acceptance %.3f"
              % (name, rng, chains, ms, sweeps, ms * 1e3 / (sweeps * n), float(np.mean(r))), flush=True)
