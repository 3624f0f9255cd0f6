"""Per-step time of the production kernel on the reference's own data sets (BASELINE configs[0] and configs[1] shapes):
southernWomen (N = 32, K = 5 + 5) and n_1000 (N = 1000, Ka = 4, Kb = 6), 256 chains, constant T = 1.  Diagnostic."""
import importlib, os, sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
pkg = importlib.import_module("bipartitesbm-mcmc_amd"); syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
G = os.path.join("tests", "golden")
for name, path, na, nb, ka, kb, eps, sweeps in (("southernWomen", "southernWomen.edgelist", 18, 14, 5, 5, 0.001, 20000),
                                                ("n_1000", "bisbm-n_1000-ka_4-kb_6.edgelist", 500, 500, 4, 6, 1.0, 2000)):
    n = na + nb
    rowptr, col = pkg.load_graph(os.path.join(G, path), n)
    lab = syn.contiguous_labels(na, nb, ka, kb)
    for chains in (256, 4096):
        m = pkg.BlockModel(lab, syn.types_vector(na, nb), ka + kb, ka, kb, eps, (rowptr, col), n_chains=chains, rng="philox", seed=1)
        m.shuffle_bisbm(); mh = pkg.MetropolisHasting()
        mh.anneal(m, pkg.constant_schedule, [1.0], sweeps * n // 10, 1 << 60)
        r = mh.anneal(m, pkg.constant_schedule, [1.0], sweeps * n, 1 << 60)
        ms, upd = m.last_sweep_timing()
        print("%-14s %5d chains: %8.1f ms for %d sweeps -> %.3f us per step per chain, %.3e updates/s, %.0f sweeps/s per chain, acceptance %.3f"
              % (name, chains, ms, sweeps, ms * 1e3 / (sweeps * n), upd / (ms / 1e3), sweeps / (ms / 1e3), float(np.mean(r))), flush=True)
