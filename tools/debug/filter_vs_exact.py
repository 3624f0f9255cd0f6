"""The two forms of the two-steps pass against each other, at scale (diagnostic; round 4).

A production launch that cannot stop early runs the pass WITHOUT the early-stop bookkeeping and the running sum; BISBM_KEEP_SUM=1
runs the form WITH them in every launch.  Both must make every decision alike: same bench graph (or a denser / hub-heavy one),
many chains, SWEEPS sweeps in both forms at constant T, a cooling call and a cold sweep; every chain's labels, rates and accepted
counts must be equal.  (Written for the dropped single-precision accept filter, profiles/r04_ab_pass_scheduling.txt F1, whose
decisions it compared with the all-double form over 1.7e10 steps; kept as the cross-check of the two forms.)
usage: filter_vs_exact.py [SWEEPS] [CHAINS] [WORKLOAD]     WORKLOAD = bench | dense | hubs
"""
import hashlib
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
B = importlib.import_module("bipartitesbm-mcmc_amd")
SYN = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
sweeps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
workload = sys.argv[3] if len(sys.argv) > 3 else "bench"
if workload == "bench":
    na = nb = 500_000; ka = kb = 32; ne = 10_000_000
elif workload == "dense":   # mean degree 100: leaves of dS up to k log E with k ~ 10, degrees up to ~150
    na = nb = 100_000; ka = kb = 32; ne = 10_000_000
else:                       # few blocks of very different sizes, low epsilon
    na = nb = 300_000; ka, kb = 17, 29; ne = 4_000_000
a, b = SYN.planted_edges(na, nb, ne, ka, kb, seed=3)
rowptr, col = B.edge_to_adj((a, b), na + nb)
n = na + nb
lab = SYN.contiguous_labels(na, nb, ka, kb)
mh = B.MetropolisHasting()
out = {}
for keep in ("0", "1"):
    os.environ["BISBM_KEEP_SUM"] = keep
    g = B.BlockModel(lab, SYN.types_vector(na, nb), ka + kb, ka, kb, 1.0 if workload != "hubs" else 0.01, (rowptr, col), n_chains=chains, rng="philox", seed=5)
    g.shuffle_bisbm()
    r1 = mh.anneal(g, "constant", [1.0], sweeps * n, 1 << 60).copy()
    r2 = mh.anneal(g, "exponential", [1.3, 1.0 - 2.0 / (4 * n)], 4 * n, 1 << 60).copy()   # cools to ~0.18: margins grow with 1 / T
    r3 = mh.anneal(g, "constant", [0.03], n, 1 << 60).copy()                               # (1 / T beyond the filter's range: exact path)
    h = [hashlib.sha256(g.get_memberships(c).tobytes()).hexdigest()[:16] for c in range(chains)]
    out[keep] = (r1, r2, r3, h, g.last_counts()[0].copy())
    print("BISBM_KEEP_SUM=%s: rates %.4f %.4f %.4f" % (keep, r1.mean(), r2.mean(), r3.mean()), flush=True)
    g.close()
a, b = out["0"], out["1"]
same = all((x == y).all() for x, y in zip(a[:3], b[:3])) and a[3] == b[3] and (a[4] == b[4]).all()
diff = [c for c in range(chains) if a[3][c] != b[3][c]]
print("%s: %d chains x (%d + 4 + 1) sweeps of %d nodes = %.2e steps per form: both forms equal: %s%s" % (
    workload, chains, sweeps, n, chains * (sweeps + 5) * n, same, "" if same else "  (chains that differ: %s)" % diff[:10]), flush=True)
sys.exit(0 if same else 1)
