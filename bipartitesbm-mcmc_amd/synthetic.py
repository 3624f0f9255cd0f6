"""Synthetic planted bipartite DC-SBM graphs for the bench and the full-size tests:
edge e: a ~ U[0,Na); planted block rho(a) = floor(a*Ka/Na); with probability p_in the b-block is
floor(rho*Kb/Ka), else uniform; b is uniform inside that block's contiguous id range.  Multi-edges
are allowed, like in the shipped n_1000 dataset."""
import numpy as np


def planted_edges(na, nb, n_edges, ka, kb, seed=1, p_in=0.8):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, na, n_edges)
    ra = a * ka // na
    sb = np.where(rng.random(n_edges) < p_in, ra * kb // ka, rng.integers(0, kb, n_edges))
    lo = (sb * nb + kb - 1) // kb
    hi = ((sb + 1) * nb + kb - 1) // kb
    b = na + lo + (rng.random(n_edges) * (hi - lo)).astype(np.int64)
    return a.astype(np.uint64), b.astype(np.uint64)


def contiguous_labels(na, nb, ka, kb):
    """Equal contiguous blocks per type (then shuffle_bisbm == the CLI's --randomize start)."""
    la = (np.arange(na, dtype=np.int64) * ka) // na
    lb = ka + (np.arange(nb, dtype=np.int64) * kb) // nb
    return np.concatenate([la, lb]).astype(np.uint32)


def types_vector(na, nb):
    return np.concatenate([np.zeros(na, dtype=np.uint32), np.ones(nb, dtype=np.uint32)])
