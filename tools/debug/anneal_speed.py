"""One annealing call on the bench graph (exponential cooling from T = 2 to T = 0.1 over 30 sweeps, 256 chains; the reference's
default mode of use), with the pass depth pinned to two, to four, and chosen per launch: kernel time and accepted fraction."""
import importlib, os, sys, numpy as np
sys.path.insert(0, ".")
pkg = importlib.import_module("bipartitesbm-mcmc_amd"); syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
na = nb = 500000; E = 10_000_000; ka = kb = 32; n = na + nb
a, b = syn.planted_edges(na, nb, E, ka, kb, seed=1); rowptr, col = pkg.edge_to_adj((a, b), n)
labels = syn.contiguous_labels(na, nb, ka, kb)
mh = pkg.MetropolisHasting()
sweeps = 30
res = {}
for pin in ("2", "4", None):
    os.environ.pop("BISBM_PASS_DEPTH", None)
    if pin: os.environ["BISBM_PASS_DEPTH"] = pin
    m = pkg.BlockModel(labels, syn.types_vector(na, nb), ka + kb, ka, kb, 1.0, (rowptr, col), n_chains=256, rng="philox", seed=1)
    m.shuffle_bisbm()
    r = mh.anneal(m, "exponential", [2.0, 0.9999999], sweeps * n, 1 << 60)
    ms, upd = m.last_sweep_timing()
    res[pin] = (r.copy(), m.get_entropy().copy())
    print("depth %-5s %8.1f ms for %d sweeps -> %.3f us per step per chain, %.4g updates/s, accepted %.3f" %
          (pin or "free", ms, sweeps, ms * 1e3 / (sweeps * n), upd / ms * 1e3, float(np.mean(r))), flush=True)
    m.close()
assert all((res[k][0] == res["2"][0]).all() and (res[k][1] == res["2"][1]).all() for k in res), "results depend on the depth"
print("rates and sums of dS bit-equal across the three runs")
