"""Debug: where do the generic compat kernel and the CPU checker part ways on wide-K graphs with a linear schedule?"""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
B = importlib.import_module("bipartitesbm-mcmc_amd")
SYN = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
BIG = 1 << 60

def graph(seed, na, nb, ne, ka, kb):
    a, b = SYN.planted_edges(na, nb, ne, ka, kb, seed=seed)
    return O.edge_to_csr(a, b, na + nb)

def run(na, nb, ne, ka, kb, mode, sched, kw, sweeps):
    rowptr, col = graph(11, na, nb, ne, ka, kb)
    labels = O.contiguous_labels(na, nb, ka, kb)
    n = na + nb
    o = O.OracleModel(rowptr, col, na, nb, ka, kb, 1.0, labels)
    types = SYN.types_vector(na, nb)
    if mode == "compat":
        o.seed_compat(5, 6)
        g = B.BlockModel(labels, types, ka + kb, ka, kb, 1.0, (rowptr, col), rng="compat", seed=5, gen_seed=6)
    else:
        o.seed_philox(777, 3)
        g = B.BlockModel(labels, types, ka + kb, ka, kb, 1.0, (rowptr, col), rng="philox", seed=777, first_chain_id=3)
    o.shuffle_bisbm(); g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    ro = o.anneal(sched, kw, sweeps * n, BIG)
    rg = mh.anneal(g, sched, kw, sweeps * n, BIG)
    lo, lg = o.memberships(), g.get_memberships(0)
    nd = int((lo != lg).sum())
    print(f"na={na} ka={ka} kb={kb} {mode} {sched}{kw} sweeps={sweeps}: rate gpu={rg} cpu={ro} labels differ={nd}", flush=True)

for mode in ("compat", "philox"):
    run(72000, 72000, 216000, 40, 33, mode, "linear", [2.0, 1e-4], 1)
    run(72000, 72000, 216000, 40, 33, mode, "linear", [2.0, 1e-5], 1)   # T stays positive much longer
    run(72000, 72000, 216000, 40, 33, mode, "constant", [0.0], 1)
    run(72000, 72000, 216000, 40, 33, mode, "constant", [-1.0], 1)
    run(7200, 7200, 21600, 40, 33, mode, "linear", [2.0, 1e-4], 3)
    run(20000, 20000, 100000, 2, 2, mode, "linear", [2.0, 1e-4], 3)
    run(72000, 72000, 216000, 4, 3, mode, "linear", [2.0, 1e-4], 1)
