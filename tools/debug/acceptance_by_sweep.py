import importlib, sys, numpy as np
sys.path.insert(0,'.')
pkg = importlib.import_module("bipartitesbm-mcmc_amd"); syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
na=nb=500000; E=10_000_000; ka=kb=32; n=na+nb
a,b = syn.planted_edges(na,nb,E,ka,kb,seed=1); rowptr,col = pkg.edge_to_adj((a,b),n)
labels = syn.contiguous_labels(na,nb,ka,kb)
m = pkg.BlockModel(labels, syn.types_vector(na,nb), ka+kb, ka, kb, 1.0, (rowptr,col), n_chains=64, rng="philox", seed=20240229)
m.shuffle_bisbm(); mh = pkg.MetropolisHasting()
for s in range(8):
    r = mh.anneal(m, pkg.constant_schedule, [1.0], n, 1<<60)
    lab = m.get_memberships(0)
    same_as_planted = (lab == labels).mean()
    print("sweep", s, "acceptance", float(np.mean(r)), "kernel ms", m.last_sweep_timing()[0], flush=True)
