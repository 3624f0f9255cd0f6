import importlib, sys, numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("bipartitesbm-mcmc_amd"); syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
na = nb = 500000; E = 10_000_000; ka = kb = 32; n = na + nb
a, b = syn.planted_edges(na, nb, E, ka, kb, seed=1); rowptr, col = pkg.edge_to_adj((a, b), n)
labels = syn.contiguous_labels(na, nb, ka, kb)
m = pkg.BlockModel(labels, syn.types_vector(na, nb), ka + kb, ka, kb, 1.0, (rowptr, col), n_chains=256, rng="philox", seed=1)
m.shuffle_bisbm(); mh = pkg.MetropolisHasting()
for name, sched, kw, dur in [("constant", pkg.constant_schedule, [1.0], n), ("constant", pkg.constant_schedule, [1.0], n),
                             ("exponential", "exponential", [1.0, 0.99999999], 3 * n), ("linear", "linear", [1.0, 1e-9], 3 * n),
                             ("logarithmic", "logarithmic", [10.0, 2.0], 3 * n), ("abrupt_cool", "abrupt_cool", [1.5e6], 3 * n),
                             ("constant T=0.5", pkg.constant_schedule, [0.5], 3 * n), ("exponential 6 sweeps", "exponential", [1.0, 0.99999999], 6 * n)]:
    r = mh.anneal(m, sched, kw, dur, dur if len(sys.argv) > 1 and sys.argv[1] == "await" else 1 << 60)  # "await": early-stop bookkeeping on
    ms, upd = m.last_sweep_timing()
    print("%-22s %8.1f ms for %d sweeps -> %.3f us per step per chain, acceptance %.3f" % (name, ms, dur // n, ms * 1e3 / (dur), float(np.mean(r))), flush=True)
