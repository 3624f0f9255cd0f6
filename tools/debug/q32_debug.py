import importlib, os, sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import oracle_lib as O
from test_gpu_parity import _random_graph, gpu_model, BIG
B = importlib.import_module("bipartitesbm-mcmc_amd")
ka, kb, eps = 17, 29, 0.5
na, nb = 24_011, 23_003
rowptr, col = _random_graph(15, na, nb, 480_000, ka, kb)
n = na + nb
planted = O.contiguous_labels(na, nb, ka, kb)
mh = B.MetropolisHasting()
runs = [("constant", [1.0], 2 * n, BIG), ("constant", [0.6], n, BIG), ("exponential", [1.5, 0.99997], 4 * n, n // 2),
        ("abrupt_cool", [1.5 * n], 3 * n, BIG), ("linear", [1.2, 1.0 / (2 * n)], 2 * n, BIG), ("constant", [1.0], n + 77, BIG)]
for start in ("randomised", "planted"):
    ms = {}
    for pin in ("4", "2", "single"):
        os.environ.pop("BISBM_SINGLE_STEPS", None); os.environ.pop("BISBM_PASS_DEPTH", None)
        if pin == "single": os.environ["BISBM_SINGLE_STEPS"] = "1"
        else: os.environ["BISBM_PASS_DEPTH"] = pin
        g = gpu_model(rowptr, col, na, nb, ka, kb, eps, planted, n_chains=5, rng="philox", seed=78, first_chain_id=1)
        g.shuffle_bisbm() if start == "randomised" else g.init_bisbm()
        ents = []
        for s, kw, dur, aw in runs:
            mh.anneal(g, s, kw, dur, aw)
            ents.append(g.get_entropy().copy())
        ms[pin] = ents
    o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, planted)
    o.seed_philox(78, 1)
    o.shuffle_bisbm() if start == "randomised" else o.init_bisbm()
    for i, (s, kw, dur, aw) in enumerate(runs):
        o.anneal(s, kw, dur, aw)
        print(start, s, kw, "4==2", (ms["4"][i] == ms["2"][i]).all(), "2==single", (ms["2"][i] == ms["single"][i]).all(),
              "diff", (ms["4"][i] - ms["2"][i]).max(), "oracle-4", o.get_entropy() - ms["4"][i][0], "oracle-2", o.get_entropy() - ms["2"][i][0])
