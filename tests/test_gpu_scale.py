"""GPU tests (-m gpu) beyond per-chain parity with the oracle:
  * the production chain's stationary distribution on the enumerable graph (chi-square against exp(-S));
  * BASELINE configs[3] ("estimate mode": 512 chains on the N = 1e6 graph, 48+48 -> 32+32 blocks through agg_merge
    stages with greedy sweeps) with a sample of chains equal to their oracle runs;
  * BASELINE configs[4]'s per-GPU shape (N = 4e6, E = 5e7, Ka = Kb = 64): size-independent properties."""
import importlib

import numpy as np
import pytest

import cases
import oracle_lib as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _every_launch_keeps_its_own_sum(monkeypatch):
    """The checks of this module compare the sum of the dS values the kernel itself computed with the change of the full
    description length; a production launch that cannot stop early would otherwise take that sum FROM the description length
    (DESIGN.md section 6), and the check would hold by construction."""
    monkeypatch.setenv("BISBM_KEEP_SUM", "1")

B = importlib.import_module("bipartitesbm-mcmc_amd")
SYN = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
BIG = 1 << 60


def gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, **kw):
    return B.BlockModel(labels, SYN.types_vector(na, nb), ka + kb, ka, kb, eps, (rowptr, col), **kw)


def assert_state_equal(g, o, chain=0):
    assert (g.get_memberships(chain) == o.memberships()).all()
    assert (g.get_m(chain) == o.m()).all()
    assert (g.get_m_r(chain) == o.m_r()).all()
    assert (g.get_n_r(chain) == o.n_r()).all()
    assert (g.get_eta_rk_(chain) == o.eta()).all()


# ------------------------------------------------------------------ stationary distribution of the production chain
@pytest.mark.parametrize("T", [1.0, 2.0, 0.6])
def test_gpu_chains_sample_exp_minus_S(T):
    """32768 independent Philox chains of the production kernel on the 6+6-node graph at constant temperature T, one sample
    each after a burn-in from a randomised start, against exp(-S / T) over the 3844 admissible states (S = entropy(),
    blockmodel.cc:753-787; the acceptance rule a = -dS / T + log(accu_r), metropolis_hasting.cc:52-57, has that stationary
    distribution) -- the same test tests/test_cross_mode.py runs on the oracle in the reference's arithmetic."""
    rowptr, col = cases.enumerable_graph()
    na, nb = cases.ENUM_NA, cases.ENUM_NB
    start = O.contiguous_labels(na, nb, 2, 2)
    chains, burn_in = 32768, 100
    g = gpu_model(rowptr, col, na, nb, 2, 2, cases.ENUM_EPS, start, n_chains=chains, rng="philox", seed=4242)
    g.shuffle_bisbm()
    B.MetropolisHasting().anneal(g, "constant", [T], burn_in * (na + nb), BIG)
    labs = [g.get_memberships(c) for c in range(chains)]
    codes = np.array([cases.state_code(l) for l in labs])
    states, _, S = cases.enumerable_states()
    target = np.exp(-(S - S.min()) / T)
    stat, dof, p = cases.chi_square(codes, states, target / target.sum())
    print("GPU philox, T = %g: chi2 = %.1f on %d dof, p = %.3g" % (T, stat, dof, p))
    assert p > 1e-3, (stat, dof, p)
    w = np.exp(-(S - S.min()) / (1.25 * T))  # power: a somewhat wrong target is rejected by the same samples
    assert cases.chi_square(codes, states, w / w.sum())[2] < 1e-6
    # and the chains are the oracle's chains
    for c in (0, 1, 777, chains - 1):
        o = O.OracleModel(rowptr, col, na, nb, 2, 2, cases.ENUM_EPS, start)
        o.seed_philox(4242, c)
        o.shuffle_bisbm()
        o.anneal("constant", [T], burn_in * (na + nb), BIG)
        assert (o.memberships() == labs[c]).all()


# ------------------------------------------------------------------ BASELINE configs[3]: estimate mode
def test_config4_estimate_mode_512_chains():
    """The N = 1e6 / E = 1e7 graph, 512 chains, an initial partition of 48 + 48 blocks merged down to 32 + 32 through
    bisbm_agg_merge stages (blockmodel.cc:109-206; driver loop mcmc_main.cc:425-444: one merge, one greedy sweep per
    stage), then a sweep at T = 1.  A sample of chains must equal their oracle runs; every chain must be consistent."""
    na = nb = 500_000
    n = na + nb
    a, b = SYN.planted_edges(na, nb, 10_000_000, 32, 32, seed=1)
    rowptr, col = B.edge_to_adj((a, b), n)
    del a, b
    k0 = 48
    labels = SYN.contiguous_labels(na, nb, k0, k0)
    chains = 512
    g = gpu_model(rowptr, col, na, nb, k0, k0, 1.0, labels, n_chains=chains, rng="philox", seed=2024, first_chain_id=100)
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    stages = [44, 40, 36, 32]
    sample = (0, 255, 511)
    oracles = []
    for c in sample:
        o = O.OracleModel(rowptr, col, na, nb, k0, k0, 1.0, labels)
        o.seed_philox(2024, 100 + c)
        o.shuffle_bisbm()
        oracles.append(o)
    prev = k0
    for k in stages:
        g.agg_merge(prev - k, prev - k, 10)
        assert (g.KA, g.KB) == (k, k)
        rates = mh.anneal(g, "abrupt_cool", [0.0], n, BIG)  # T = 0: greedy sweep
        for o, c in zip(oracles, sample):
            assert o.agg_merge(prev - k, prev - k, 10) == 0
            ro = o.anneal("abrupt_cool", [0.0], n, BIG)
            assert ro == rates[c], (k, c, ro, rates[c])
        prev = k
    for o, c in zip(oracles, sample):
        assert_state_equal(g, o, c)
    ent = g.entropy()
    for o, c in zip(oracles, sample):
        assert ent[c] == pytest.approx(o.entropy(), rel=1e-9)
    # every chain: 32 + 32 non-empty blocks, counts consistent with the labels
    for c in range(0, chains, 37):
        lab = g.get_memberships(c)
        n_r = g.get_n_r(c)
        assert (np.bincount(lab, minlength=64) == n_r).all() and n_r.min() >= 1
        assert lab[:na].max() == 31 and lab[na:].min() == 32 and lab[na:].max() == 63
        assert g.get_m_r(c).sum() == 2 * 10_000_000
    # the chains go on at T = 1 in the production kernel's K <= 32 variant
    s0 = g.entropy()
    cum0 = g.get_entropy()
    rates = mh.anneal(g, "constant", [1.0], n, BIG)
    for o, c in zip(oracles, sample):
        assert o.anneal("constant", [1.0], n, BIG) == rates[c]
        assert (o.memberships() == g.get_memberships(c)).all()
    s1 = g.entropy()
    dcum = g.get_entropy() - cum0  # accepted dS of this call == change of the description length
    assert np.allclose(s1 - s0, dcum, rtol=1e-9, atol=1e-6 * np.abs(dcum).max())
    assert ((rates > 0.05) & (rates <= 1.0)).all()


def test_estimate_mode_with_the_one_argument_merge_at_full_size():
    """The same graph, 128 chains from 40 + 40 blocks, two one-argument merges (blockmodel.cc:208-271: every chain decides
    which type loses blocks, so the chains end in several shapes and the handle regroups them) with greedy sweeps, then a
    sweep at T = 1 with all groups in flight together.  Sampled chains equal their oracle runs; every chain consistent."""
    na = nb = 500_000
    n = na + nb
    a, b = SYN.planted_edges(na, nb, 10_000_000, 32, 32, seed=1)
    rowptr, col = B.edge_to_adj((a, b), n)
    del a, b
    k0 = 40
    labels = SYN.contiguous_labels(na, nb, k0, k0)
    chains = 128
    g = gpu_model(rowptr, col, na, nb, k0, k0, 1.0, labels, n_chains=chains, rng="philox", seed=77)
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    mh.anneal(g, "constant", [1.0], n, BIG)
    sample = (0, 63, 127)
    oracles = []
    for c in sample:
        o = O.OracleModel(rowptr, col, na, nb, k0, k0, 1.0, labels)
        o.seed_philox(77, c)
        o.shuffle_bisbm()
        o.anneal("constant", [1.0], n, BIG)
        oracles.append(o)
    for diff in (9, 7):
        g.agg_merge(diff, None, 10)
        rates = mh.anneal(g, "abrupt_cool", [0.0], n, BIG)
        for o, c in zip(oracles, sample):
            assert o.agg_merge_total(diff, 10) == 0
            assert o.anneal("abrupt_cool", [0.0], n, BIG) == rates[c]
            assert g.ka_kb(c) == (o.ka, o.kb)
    shapes = {g.ka_kb(c) for c in range(chains)}
    print("shapes after two one-argument merges:", sorted(shapes))
    assert len(shapes) > 1 and all(ka + kb == 2 * k0 - 16 for ka, kb in shapes)
    s0, cum0 = g.entropy(), g.get_entropy()
    rates = mh.anneal(g, "constant", [1.0], n, BIG)
    for o, c in zip(oracles, sample):
        assert o.anneal("constant", [1.0], n, BIG) == rates[c]
        assert_state_equal(g, o, c)
    s1 = g.entropy()
    dcum = g.get_entropy() - cum0
    assert np.allclose(s1 - s0, dcum, rtol=1e-9, atol=1e-6 * np.abs(dcum).max())
    for c in range(0, chains, 13):
        ka, kb = g.ka_kb(c)
        lab, n_r = g.get_memberships(c), g.get_n_r(c)
        assert (np.bincount(lab, minlength=ka + kb) == n_r).all() and n_r.min() >= 1
        assert lab[:na].max() == ka - 1 and lab[na:].min() == ka and lab[na:].max() == ka + kb - 1
        assert g.get_m_r(c).sum() == 2 * 10_000_000
    ms, updates = g.last_sweep_timing()
    assert updates == chains * n and ms > 0


@pytest.mark.parametrize("k", [16, 8])
def test_deep_passes_at_full_size(k, monkeypatch):
    """The N = 1e6 / E = 1e7 graph with 16 + 16 and 8 + 8 blocks, 64 chains: the four- and eight-steps passes pinned (the depth
    is otherwise chosen from timed launches), from the planted partition (few steps move: long passes commit) and from a
    randomised start (nearly every step moves: passes are cut short all the time).  Sampled chains equal their oracle runs;
    every chain's incremental state equals a recount and sum dS the change of the description length."""
    na = nb = 500_000
    n = na + nb
    a, b = SYN.planted_edges(na, nb, 10_000_000, k, k, seed=1)
    rowptr, col = B.edge_to_adj((a, b), n)
    del a, b
    planted = SYN.contiguous_labels(na, nb, k, k)
    chains = 64
    mh = B.MetropolisHasting()
    monkeypatch.setenv("BISBM_PASS_DEPTH", "8" if k == 8 else "4")
    for start in ("planted", "randomised"):
        g = gpu_model(rowptr, col, na, nb, k, k, 1.0, planted, n_chains=chains, rng="philox", seed=5)
        oracles = []
        for c in (0, 63):
            o = O.OracleModel(rowptr, col, na, nb, k, k, 1.0, planted)
            o.seed_philox(5, c)
            oracles.append((c, o))
        if start == "planted":
            g.init_bisbm()
            for _, o in oracles:
                o.init_bisbm()
        else:
            g.shuffle_bisbm()
            for _, o in oracles:
                o.shuffle_bisbm()
        s0, cum0 = g.entropy(), g.get_entropy()
        rates = mh.anneal(g, "constant", [1.0], 2 * n, BIG)
        for c, o in oracles:
            assert o.anneal("constant", [1.0], 2 * n, BIG) == rates[c], (start, c)
            assert_state_equal(g, o, c)
        dcum = g.get_entropy() - cum0
        assert np.allclose(g.entropy() - s0, dcum, rtol=1e-9, atol=1e-6 * max(1.0, np.abs(dcum).max()))
        state = [(c, g.get_m(c), g.get_m_r(c), g.get_n_r(c), g.get_eta_rk_(c)) for c in range(0, chains, 9)]
        g.init_bisbm()  # recount from the labels
        for c, m, m_r, n_r, eta in state:
            assert (g.get_m(c) == m).all() and (g.get_m_r(c) == m_r).all() and (g.get_n_r(c) == n_r).all()
            assert (g.get_eta_rk_(c) == eta).all() and n_r.sum() == n and m_r.sum() == 2 * 10_000_000


# ------------------------------------------------------------------ BASELINE configs[4]: per-GPU shape
@pytest.mark.parametrize("chains,container", [(256, 0), (1024, 1), (8192, 8)])
def test_config5_shape_properties(chains, container, monkeypatch, capfd):
    """N_a = N_b = 2e6, E = 5e7, Ka = Kb = 64 (the K > 32 variant of the production kernel: two steps per pass with two
    blocks per lane, a window of eta in LDS) -- the per-GPU shape of BASELINE configs[4], at a quarter of its chains on a plain
    handle, at its full per-GPU load of 1024 chains behind a multi-device handle over this one device (`devices=[0]`: what
    each GPU of the 8-GPU configuration runs, pooling of the marginals through RCCL included), and at the configuration's own
    chain count -- 8192 chains behind a handle of EIGHT device entries, 1024 chains each, all eight on this one GPU (the
    sharding, the per-entry engines side by side and the pooled marginals over eight 1 GB histograms as the 8-GPU node would
    run them, the exchange as peer copies because the entries are one device): a sweep at constant T and a
    sweep under a cooling schedule keep the incremental state equal to a recount, block sizes sum to N, and the sum of accepted
    dS equals the change of the full description length, in every chain."""
    na = nb = 2_000_000
    ka = kb = 64
    E = 50_000_000
    a, b = SYN.planted_edges(na, nb, E, ka, kb, seed=1)
    rowptr, col = B.edge_to_adj((a, b), na + nb)
    del a, b
    labels = SYN.contiguous_labels(na, nb, ka, kb)
    monkeypatch.setenv("BISBM_POOL_LOG", "1")
    g = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, labels, n_chains=chains, rng="philox", seed=5, **({"devices": [0] * container} if container else {}))
    g.shuffle_bisbm()
    s0 = g.entropy()
    rates = B.MetropolisHasting().anneal(g, "constant", [1.0], na + nb, BIG)
    assert ((rates > 0.3) & (rates <= 1.0)).all()
    s1 = g.entropy()
    cum = g.get_entropy()
    assert np.allclose(s1 - s0, cum, rtol=1e-9, atol=1e-6 * np.abs(cum).max())
    picks = (0, 7, chains - 1)
    labs_after_first = [g.get_memberships(c) for c in picks]
    rates = B.MetropolisHasting().anneal(g, "exponential", [1.5, 0.9999999], na + nb, BIG)  # (the cooling-schedule variant)
    assert ((rates > 0.2) & (rates <= 1.0)).all()
    s2 = g.entropy()
    cum2 = g.get_entropy()
    assert np.allclose(s2 - s0, cum2, rtol=1e-9, atol=1e-6 * np.abs(cum2).max())
    before = [(g.get_m(c), g.get_m_r(c), g.get_n_r(c), g.get_eta_rk_(c)) for c in picks]
    labs = [g.get_memberships(c) for c in picks]
    n_r_all = [g.get_n_r(c) for c in range(0, chains, 17)]
    assert all(x.sum() == na + nb and (x > 0).all() for x in n_r_all)
    g.init_bisbm()  # recount from the labels
    for (m, m_r, n_r, eta), c, lab in zip(before, picks, labs):
        assert (g.get_m(c) == m).all() and (g.get_m_r(c) == m_r).all()
        assert (g.get_n_r(c) == n_r).all() and (g.get_eta_rk_(c) == eta).all()
        assert n_r.sum() == na + nb and m_r.sum() == 2 * E
        assert (np.bincount(lab, minlength=ka + kb) == n_r).all()
    assert all((g.get_n_r(c) == x).all() for c, x in zip(range(0, chains, 17), n_r_all))
    if container:
        # configs[4]'s exchange at this device's share of it: one sample of every chain into the histogram (1 GB of counters),
        # pooled by reduce-scatter -> argmax -> all-gather through RCCL (one rank here), against numpy on the host copy
        g.marginals_reset()
        g.marginals_accumulate(None)
        capfd.readouterr()
        lab_map = g.marginals_map()
        log = capfd.readouterr().err
        if container == 1:
            assert "[bisbm pool] RCCL path" in log and "peer-copy" not in log, log
        else:
            assert "[bisbm pool] peer-copy path" in log and "RCCL path" not in log, log
        counts = g.marginals_get()
        assert counts.shape == (na + nb, 64) and int(counts.sum()) == chains * (na + nb)
        want = counts.argmax(axis=1) + np.where(np.arange(na + nb) >= na, ka, 0)
        assert (lab_map == want).all()
        del counts, want
    # chains are distinct (keyed by chain id) and reproducible: chain 7 re-run alone gives the same labels
    assert (labs[0] != labs[1]).any()
    solo = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, labels, n_chains=1, rng="philox", seed=5, first_chain_id=7)
    solo.shuffle_bisbm()
    B.MetropolisHasting().anneal(solo, "constant", [1.0], na + nb, BIG)
    assert (solo.get_memberships(0) == labs_after_first[1]).all()
