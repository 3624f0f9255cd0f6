"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes), against
  * the golden reference outputs recorded by the survey (mt19937-compat mode, bit-exact),
  * the oracle restatement on the same seeded inputs (both RNG modes, integers bit-exact),
  * size-independent properties at BASELINE.json's full size."""
import ctypes as C
import importlib
import math
import os

import numpy as np
import pytest

import cases
import oracle_lib as O

pytestmark = pytest.mark.gpu

B = importlib.import_module("bipartitesbm-mcmc_amd")
SYN = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
BIG = 1 << 60


def gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, **kw):
    return B.BlockModel(labels, SYN.types_vector(na, nb), ka + kb, ka, kb, eps, (rowptr, col), **kw)


def sum_dS_close(got, o, rel=1e-9):
    """The running sum of accepted dS against the oracle's: 1e-9 relative (north_star), plus what a DIFFERENCE of description
    lengths can resolve -- a production launch without the early-stop bookkeeping advances the sum by the change of the
    description length S over the call, so near the mode, where the sum hovers around zero, its error is
    a few ulps of S (1e-12 |S| allowed), not of the sum."""
    want = o.get_entropy()
    return abs(got - want) <= rel * abs(want) + 1e-12 * abs(o.entropy())


def assert_state_equal(g, o, chain=0):
    assert (g.get_memberships(chain) == o.memberships()).all()
    assert (g.get_m(chain) == o.m()).all()
    assert (g.get_m_r(chain) == o.m_r()).all()
    assert (g.get_n_r(chain) == o.n_r()).all()
    assert (g.get_eta_rk_(chain) == o.eta()).all()


# ------------------------------------------------------------------ golden: reference outputs
def _golden_run(golden, key):
    g = golden["compat_rng"][key]
    rowptr, col, na, nb = O.load_graph(g["graph"])
    labels = O.contiguous_labels(na, nb, g["ka"], g["kb"])
    m = gpu_model(rowptr, col, na, nb, g["ka"], g["kb"], g["epsilon"], labels, rng="compat", seed=42, gen_seed=43)
    m.shuffle_bisbm()
    return g, m


@pytest.mark.parametrize("key", ["sample_southernWomen", "sample_n_1000"])
def test_golden_first_sweep(golden, key):
    g, m = _golden_run(golden, key)
    assert m.entropy()[0] == pytest.approx(g["S0"], rel=1e-13)
    rate = B.MetropolisHasting().anneal(m, g["schedule"], g["kwargs"], g["duration"], BIG)
    assert rate == g["rate"]
    assert m.get_entropy()[0] == pytest.approx(g["sum_dS"], rel=1e-12)


def test_golden_scenario1_config1(golden):
    """BASELINE config 1: southernWomen, Ka=Kb=5, exponential(10, 0.1), early stop; labels bit-exact."""
    g, m = _golden_run(golden, "scenario1_config1")
    rate = B.MetropolisHasting().anneal(m, B.exponential_schedule, g["kwargs"], g["duration"], g["steps_await"])
    assert rate == g["rate"]
    acc, sw = m.last_counts()
    assert acc[0] == g["accepted"] and sw[0] == g["sweeps"]
    assert list(m.get_memberships()) == g["labels"]
    assert m.get_entropy()[0] == pytest.approx(g["sum_dS"], rel=1e-12)
    assert m.entropy()[0] == pytest.approx(g["entropy_approx"], abs=1e-3)


def test_golden_scenario2(golden):
    g, m = _golden_run(golden, "scenario2_n1000_constant")
    rate = B.MetropolisHasting().anneal(m, g["schedule"], g["kwargs"], g["duration"], BIG)
    assert rate == g["rate"]
    assert m.get_entropy()[0] == pytest.approx(g["sum_dS"], rel=1e-12)


def test_golden_scenario3(golden):
    g, m = _golden_run(golden, "scenario3_n1000_abrupt")
    rate = B.MetropolisHasting().anneal(m, g["schedule"], g["kwargs"], g["duration"], g["steps_await"])
    assert rate == g["rate"]
    assert m.get_entropy()[0] == pytest.approx(g["sum_dS"], rel=1e-12)


def test_golden_rng_free_state(golden):
    g = golden["rng_free"]["n_1000"]
    rowptr, col, na, nb = O.load_graph("n_1000")
    labels = O.load_memberships(os.path.join(O.GOLDEN, g["membership"]))
    m = gpu_model(rowptr, col, na, nb, 4, 6, 1.0, labels)
    m.init_bisbm()
    assert list(m.get_m_r()) == g["m_r"]
    assert list(m.get_m()[0]) == g["m_row0"]
    assert m.entropy()[0] == pytest.approx(g["entropy"], rel=1e-13)
    g = golden["rng_free"]["southernWomen"]
    rowptr, col, na, nb = O.load_graph("southernWomen")
    m = gpu_model(rowptr, col, na, nb, 5, 5, 0.001, O.labels_from_sizes(g["block_sizes"]))
    m.init_bisbm()
    assert list(m.get_m_r()) == g["m_r"]
    assert m.entropy()[0] == pytest.approx(g["entropy"], rel=1e-13)


# ------------------------------------------------------------------ oracle, same seeded inputs
_random_graph = cases.random_graph
CASES = cases.CASES


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
@pytest.mark.parametrize("mode", ["compat", "philox"])
def test_matches_oracle(case, mode):
    name, na, nb, ne, ka, kb, eps, hubs, iso = case
    rowptr, col = _random_graph(11, na, nb, ne, ka, kb, hubs, iso)
    labels = O.contiguous_labels(na, nb, ka, kb)
    n = na + nb
    o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
    if mode == "compat":
        o.seed_compat(5, 6)
        g = gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, rng="compat", seed=5, gen_seed=6)
    else:
        o.seed_philox(777, 3)
        g = gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, rng="philox", seed=777, first_chain_id=3)
    o.shuffle_bisbm()
    g.shuffle_bisbm()
    assert_state_equal(g, o)
    assert g.entropy()[0] == pytest.approx(o.entropy(), rel=1e-12)
    mh = B.MetropolisHasting()
    # (the linear ramp ends at T = 0.5: below zero the reference accepts cross-type proposals -- a = -dS/T = +inf,
    # metropolis_hasting.cc:55-57 -- and corrupts its own state; the engine never does, see DESIGN.md "hazards")
    for sched, kw, dur, await_ in [("constant", [1.0], 6 * n, BIG), ("linear", [2.0, 1.5 / (3 * n)], 3 * n, BIG),
                                   ("abrupt_cool", [float(n)], 4 * n, BIG), ("exponential", [3.0, 0.999], 3 * n, 2 * n),
                                   ("logarithmic", [1.0, 2.0], 2 * n, BIG),
                                   # T underflows to 0 inside the call; on the way 1 / T overflows for some fifty steps, where
                                   # -1 / T * dS is +-inf or (r == s: dS = 0) NaN: accepted iff dS < 0 (metropolis_hasting.cc:54-59)
                                   ("exponential", [1e-3, 0.5], 2 * n, BIG)]:
        ro = o.anneal(sched, kw, dur, await_)
        rg = mh.anneal(g, sched, kw, dur, await_)
        assert rg == ro, (sched, rg, ro)
        assert_state_equal(g, o)
        acc, sw = g.last_counts()
        assert acc[0] == o.last_accepted and sw[0] == o.last_sweeps
        assert sum_dS_close(g.get_entropy()[0], o) or abs(g.get_entropy()[0] - o.get_entropy()) <= 1e-9
    assert g.entropy()[0] == pytest.approx(o.entropy(), rel=1e-9)


@pytest.mark.parametrize("window", ["3", "1", None])
@pytest.mark.parametrize("name", ["k32_eta_in_hbm", "k64_eta_in_hbm", "k8_eta_in_hbm"])
def test_eta_window_in_lds(name, window, monkeypatch):
    """Where eta[K][max degree + 1] does not fit beside the rest of a chain's state in LDS (here: one hub of degree 600), the
    production kernel keeps a window of it there -- the rows of the phase's own type, a run of consecutive degrees placed where
    most nodes are -- and steps of nodes of other degrees take the general step with eta in HBM.  Whatever the window (the
    library's choice, or three degrees / one degree wide: most steps outside), the chains equal the oracle's."""
    if window is None:
        monkeypatch.delenv("BISBM_ETA_WINDOW", raising=False)
    else:
        monkeypatch.setenv("BISBM_ETA_WINDOW", window)
    _, na, nb, ne, ka, kb, eps, hubs, iso = cases.CASE[name]
    rowptr, col = _random_graph(11, na, nb, ne, ka, kb, hubs, iso)
    labels = O.contiguous_labels(na, nb, ka, kb)
    n, chains, first = na + nb, 3, 11
    g = gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, n_chains=chains, rng="philox", seed=31, first_chain_id=first)
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    runs = [("constant", [1.0], 3 * n, BIG), ("exponential", [1.5, 0.9999], 6 * n, n), ("abrupt_cool", [1.5 * n], 3 * n, BIG)]
    got = [mh.anneal(g, s_, kw, dur, aw).copy() for s_, kw, dur, aw in runs]
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
        o.seed_philox(31, first + c)
        o.shuffle_bisbm()
        for (s_, kw, dur, aw), rates in zip(runs, got):
            assert o.anneal(s_, kw, dur, aw) == rates[c], (s_, c)
        assert_state_equal(g, o, c)
        assert sum_dS_close(g.get_entropy()[c], o)


def test_hot_step_many_chains_philox():
    """The production kernel's hot step (closed-form log_q tier) with several chains per launch: every chain equals
    its own oracle run (chain ids key the streams, workgroup <-> chain mapping, per-chain state in LDS)."""
    name, na, nb, ne, ka, kb, eps, hubs, iso = next(c for c in CASES if c[0] == "direct_tier")
    rowptr, col = _random_graph(11, na, nb, ne, ka, kb, hubs, iso)
    labels = O.contiguous_labels(na, nb, ka, kb)
    n, chains, first = na + nb, 12, 40
    g = gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, n_chains=chains, rng="philox", seed=5150, first_chain_id=first)
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    rates = mh.anneal(g, "constant", [1.0], 2 * n, BIG)
    rates2 = mh.anneal(g, "exponential", [2.0, 0.99995], 2 * n, BIG)
    cum = g.get_entropy()
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
        o.seed_philox(5150, first + c)
        o.shuffle_bisbm()
        assert o.anneal("constant", [1.0], 2 * n, BIG) == rates[c]
        assert o.anneal("exponential", [2.0, 0.99995], 2 * n, BIG) == rates2[c]
        assert_state_equal(g, o, c)
        assert sum_dS_close(cum[c], o)


@pytest.mark.parametrize("ka,kb", [(32, 32), (64, 64), (33, 57)])
def test_two_steps_per_pass_equals_the_serial_chain(ka, kb, monkeypatch):
    """(64 + 64 and 33 + 57 blocks: the variant for more than 32 blocks of a type, whose lanes hold two blocks each --
    step_pair64 -- under the same checks, plus cooling schedules with the early stop armed.)
    K = 32 + 32 at constant T: the production kernel evaluates steps q and q+1 in the two halves of the wave and
    commits both when step q provably left step q+1's inputs alone.  Many blocks make that the
    common case (with K = 2 + 2 nearly every pair clashes), so this is the test of the commit-both path: chains equal
    their oracle runs sweep by sweep -- from a randomised start (most steps move) and from the planted partition (most
    proposals are r == s) -- and equal the same kernel forced to one step per pass."""
    na = nb = 24_000
    rowptr, col = _random_graph(5, na, nb, 480_000, ka, kb)
    n = na + nb
    planted = O.contiguous_labels(na, nb, ka, kb)
    mh = B.MetropolisHasting()
    chains = 6
    for start in ("randomised", "planted"):
        g = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, planted, n_chains=chains, rng="philox", seed=77, first_chain_id=2)
        monkeypatch.setenv("BISBM_SINGLE_STEPS", "1")
        h = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, planted, n_chains=chains, rng="philox", seed=77, first_chain_id=2)
        if start == "randomised":
            h.shuffle_bisbm()
        else:
            h.init_bisbm()
        single = [mh.anneal(h, "constant", [1.0], n, BIG).copy() for _ in range(3)]
        monkeypatch.delenv("BISBM_SINGLE_STEPS")
        oracles = []
        for c in (0, chains - 1):
            o = O.OracleModel(rowptr, col, na, nb, ka, kb, 1.0, planted)
            o.seed_philox(77, 2 + c)
            oracles.append((c, o))
        if start == "randomised":
            g.shuffle_bisbm()
            for _, o in oracles:
                o.shuffle_bisbm()
        else:
            g.init_bisbm()
            for _, o in oracles:
                o.init_bisbm()
        for sweep in range(3):
            rates = mh.anneal(g, "constant", [1.0], n, BIG)
            assert (rates == single[sweep]).all()
            for c, o in oracles:
                assert o.anneal("constant", [1.0], n, BIG) == rates[c], (start, sweep, c)
                assert_state_equal(g, o, c)
        for c in range(chains):
            assert (g.get_memberships(c) == h.get_memberships(c)).all()
        assert np.allclose(g.get_entropy(), h.get_entropy(), rtol=0, atol=0)  # same sum, same order of additions
        for c, o in oracles:
            assert sum_dS_close(g.get_entropy()[c], o)
        if start == "planted":  # cooling schedules, the early stop armed, the greedy tail (T = 0), on the same chains
            for sched, kw, dur, aw in [("exponential", [1.5, 0.99997], 4 * n, n // 2), ("abrupt_cool", [1.5 * n], 3 * n, BIG),
                                       ("linear", [1.2, 1.0 / (2 * n)], 2 * n, BIG)]:
                rates = mh.anneal(g, sched, kw, dur, aw)
                for c, o in oracles:
                    assert o.anneal(sched, kw, dur, aw) == rates[c], (sched, c)
                    assert_state_equal(g, o, c)
    if (ka, kb) != (32, 32):
        return
    # a temperature other than 1 and a chunk that ends on an odd step (n_own not a multiple of 64)
    rowptr, col = _random_graph(6, 1003, 777, 30_000, 7, 5)
    lab = O.contiguous_labels(1003, 777, 7, 5)
    g = gpu_model(rowptr, col, 1003, 777, 7, 5, 0.5, lab, n_chains=3, rng="philox", seed=9)
    g.shuffle_bisbm()
    o = O.OracleModel(rowptr, col, 1003, 777, 7, 5, 0.5, lab)
    o.seed_philox(9, 1)
    o.shuffle_bisbm()
    for T in (2.5, 0.7):
        assert mh.anneal(g, "constant", [T], 4 * 1780, BIG)[1] == o.anneal("constant", [T], 4 * 1780, BIG)
        assert_state_equal(g, o, 1)


def test_four_and_eight_steps_per_pass_equal_the_serial_chain(monkeypatch):
    """Both block counts <= 16 (<= 8): the production kernel evaluates steps q .. q+3 (q+7) in the four rows (eight groups of
    eight lanes) of the wave and commits them in order as long as none of the earlier movers touched what the next one read
    (DESIGN.md section 6).  Chains equal their oracle runs sweep by sweep and equal the same kernel held to four, two and
    one step per pass -- on graphs with m_r > 10^4 (closed-form log_q tiers; 12 + 9 blocks: four per pass, 5 + 7: eight), on
    the n_1000 data set (4 + 6 blocks, table tier: eight per pass), under cooling schedules with the early stop armed, and
    with chunks that end mid-pass."""
    mh = B.MetropolisHasting()
    n1000 = O.load_graph("n_1000")
    big = _random_graph(8, 20_011, 17_003, 300_000, 12, 9)
    big8 = _random_graph(9, 20_011, 17_003, 300_000, 5, 7)
    for (rowptr, col, na, nb, ka, kb, eps), runs in (
            ((big[0], big[1], 20_011, 17_003, 12, 9, 1.0), [("constant", [1.0], 2, BIG), ("constant", [0.6], 1, BIG)]),
            ((big8[0], big8[1], 20_011, 17_003, 5, 7, 0.5), [("constant", [1.0], 2, BIG), ("abrupt_cool", [50_000.0], 2, BIG)]),
            ((n1000[0], n1000[1], 500, 500, 4, 6, 1.0), [("constant", [1.0], 20, BIG), ("exponential", [3.0, 0.9995], 10, 1500),
                                                         ("abrupt_cool", [2600.0], 4, BIG), ("linear", [2.0, 1e-4], 6, BIG)])):
        n = na + nb
        lab = O.contiguous_labels(na, nb, ka, kb)
        chains = 5
        models = {}
        for env in ("1", "2", "4", None):  # one step per pass, two, at most four, all the variant allows
            if env is None:
                monkeypatch.delenv("BISBM_SINGLE_STEPS", raising=False)
            else:
                monkeypatch.setenv("BISBM_SINGLE_STEPS", env)
            m = gpu_model(rowptr, col, na, nb, ka, kb, eps, lab, n_chains=chains, rng="philox", seed=321)
            m.shuffle_bisbm()
            models[env] = (m, [np.atleast_1d(mh.anneal(m, s, kw, sweeps * n, aw)).copy() for s, kw, sweeps, aw in runs])
        monkeypatch.delenv("BISBM_SINGLE_STEPS", raising=False)
        g, rates = models[None]
        for env in ("1", "2", "4"):
            h, rates_h = models[env]
            for a, b in zip(rates, rates_h):
                assert (a == b).all()
            for c in range(chains):
                assert (g.get_memberships(c) == h.get_memberships(c)).all()
            assert (g.get_entropy() == h.get_entropy()).all()  # same sum, same order of additions
        for c in (0, chains - 1):
            o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, lab)
            o.seed_philox(321, c)
            o.shuffle_bisbm()
            for (s, kw, sweeps, aw), r in zip(runs, rates):
                assert o.anneal(s, kw, sweeps * n, aw) == r[c], (s, c)
            assert_state_equal(g, o, c)
            assert sum_dS_close(g.get_entropy()[c], o)


@pytest.mark.parametrize("ka,kb,eps", [(32, 32, 1.0), (17, 29, 0.5), (24, 7, 1.0)])
def test_four_steps_per_pass_with_two_blocks_per_lane(ka, kb, eps, monkeypatch):
    """17..32 blocks of a type: a launch whose pass depth is four runs step_quad32 -- four steps in the four 16-lane rows of
    the wave, every lane holding two blocks.  The chains equal their oracle runs and the same chains held
    to two steps and to one step per pass: from a randomised start (most steps move, most followers clash) and from the planted
    partition (most proposals are r == s, passes commit all four), at constant temperatures and under cooling schedules with
    the early stop armed and the greedy tail (T = 0), with chunks that end mid-pass (class sizes not multiples of 64) -- and the
    same calls run as launches of single sweeps with the depth chosen per launch (any schedule, early stop armed or not: what
    anneal() carries between sweeps travels in the chain's scalars)."""
    na, nb = 24_011, 23_003
    rowptr, col = _random_graph(15, na, nb, 480_000, ka, kb)
    n = na + nb
    planted = O.contiguous_labels(na, nb, ka, kb)
    mh = B.MetropolisHasting()
    chains = 5
    runs = [("constant", [1.0], 2 * n, BIG), ("constant", [0.6], n, BIG), ("exponential", [1.5, 0.99997], 4 * n, n // 2),
            ("abrupt_cool", [1.5 * n], 3 * n, BIG), ("linear", [1.2, 1.0 / (2 * n)], 2 * n, BIG), ("constant", [1.0], n + 77, BIG)]
    for start in ("randomised", "planted"):
        out = {}
        for pin in ("4", "2", "single", "free"):
            monkeypatch.delenv("BISBM_SINGLE_STEPS", raising=False)
            monkeypatch.delenv("BISBM_PASS_DEPTH", raising=False)
            monkeypatch.delenv("BISBM_LAUNCH_STEPS", raising=False)
            if pin == "single":
                monkeypatch.setenv("BISBM_SINGLE_STEPS", "1")
            elif pin == "free":  # every call cut into launches of one sweep, the depth chosen per launch from the timings
                monkeypatch.setenv("BISBM_LAUNCH_STEPS", "1")
            else:
                monkeypatch.setenv("BISBM_PASS_DEPTH", pin)
            g = gpu_model(rowptr, col, na, nb, ka, kb, eps, planted, n_chains=chains, rng="philox", seed=78, first_chain_id=1)
            g.shuffle_bisbm() if start == "randomised" else g.init_bisbm()
            rates = [np.atleast_1d(mh.anneal(g, s, kw, dur, aw)).copy() for s, kw, dur, aw in runs]
            out[pin] = (g, rates, g.last_counts())
        monkeypatch.delenv("BISBM_SINGLE_STEPS", raising=False)
        monkeypatch.delenv("BISBM_PASS_DEPTH", raising=False)
        monkeypatch.delenv("BISBM_LAUNCH_STEPS", raising=False)
        g, rates, counts = out["4"]
        for pin in ("2", "single", "free"):
            h, rates_h, counts_h = out[pin]
            for a, b in zip(rates, rates_h):
                assert (a == b).all(), (start, pin)
            assert all((x == y).all() for x, y in zip(counts, counts_h))
            for c in range(chains):
                assert (g.get_memberships(c) == h.get_memberships(c)).all(), (start, pin, c)
            assert (g.get_entropy() == h.get_entropy()).all()  # same sum, same order of additions
        for c in (0, chains - 1):
            o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, planted)
            o.seed_philox(78, 1 + c)
            o.shuffle_bisbm() if start == "randomised" else o.init_bisbm()
            for (s, kw, dur, aw), r in zip(runs, rates):
                assert o.anneal(s, kw, dur, aw) == r[c], (start, s, c)
            assert_state_equal(g, o, c)
            assert sum_dS_close(g.get_entropy()[c], o)


def test_pass_depth_is_chosen_from_timed_launches_and_never_changes_results(monkeypatch, capfd):
    """With few blocks the depth of the passes (two / four / eight steps) is chosen per launch from the measured speed of the
    launches, and a long constant-temperature call runs as several launches so that the choice can follow the chain.  The
    chain does not depend on any of it: pinned depths and the free choice give bit-equal labels, rates and sums; the log shows
    that the free run measured the deepest pass and its neighbour and looked at a neighbour again later."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    n = na + nb
    lab = O.contiguous_labels(na, nb, 4, 6)
    mh = B.MetropolisHasting()
    out = {}
    for pin in ("2", "4", "8", None):
        if pin is None:
            monkeypatch.delenv("BISBM_PASS_DEPTH", raising=False)
            monkeypatch.setenv("BISBM_PASS_LOG", "1")
        else:
            monkeypatch.setenv("BISBM_PASS_DEPTH", pin)
        g = gpu_model(rowptr, col, na, nb, 4, 6, 1.0, lab, n_chains=7, rng="philox", seed=99)
        g.shuffle_bisbm()
        capfd.readouterr()
        rates = [mh.anneal(g, "constant", [1.0], 1234 * n + 77, BIG).copy(),  # (not a whole number of sweeps)
                 mh.anneal(g, "constant", [0.5], 300 * n, BIG).copy(),        # T < 1, the early stop out of reach
                 mh.anneal(g, "constant", [0.5], 300 * n, 150 * n).copy(),    # ... in reach: one launch
                 mh.anneal(g, "exponential", [2.0, 0.99999], 200 * n, BIG).copy(),
                 # steps_await = 0 at T >= 1: `u = 0 >= 0` after the FIRST sweep (metropolis_hasting.cc:96-98): one sweep, rate
                 # accepted / N, however long the call -- and steps_await = 5 (< N): u stays 0 at T = 1, the call runs to its end
                 mh.anneal(g, "constant", [1.0], 500 * n, 0).copy()]
        assert (g.last_counts()[1] == 1).all() and (g.last_counts()[0] == np.round(rates[4] * n)).all()
        rates.append(mh.anneal(g, "constant", [1.0], 450 * n, 5).copy())
        assert (g.last_counts()[1] == 450).all()
        log = capfd.readouterr().err
        out[pin] = (rates, [g.get_memberships(c) for c in range(7)], g.get_entropy(), g.last_counts())
        if pin is None:
            ran = [int(line.split("depth ")[1][0]) for line in log.splitlines() if line.startswith("[bisbm passes]")]
            # (bisbm_pass_policy.hpp: the deepest first on a graph of this size, then its neighbour; the shallowest only if four
            # steps per pass beat eight; in steady state one launch in sixteen looks at a neighbour)
            assert ran[:2] == [3, 3] and {2, 3} <= set(ran), log[-2000:]
            assert "(a look)" in log
    monkeypatch.delenv("BISBM_PASS_LOG", raising=False)
    ref = out[None]
    for pin in ("2", "4", "8"):
        for a, b in zip(ref[0], out[pin][0]):
            assert (a == b).all()
        for a, b in zip(ref[1], out[pin][1]):
            assert (a == b).all()
        assert (ref[2] == out[pin][2]).all()
        assert (ref[3][0] == out[pin][3][0]).all() and (ref[3][1] == out[pin][3][1]).all()
    o = O.OracleModel(rowptr, col, na, nb, 4, 6, 1.0, lab)
    o.seed_philox(99, 3)
    o.shuffle_bisbm()
    assert o.anneal("constant", [1.0], 1234 * n + 77, BIG) == ref[0][0][3]
    assert o.anneal("constant", [0.5], 300 * n, BIG) == ref[0][1][3]
    assert o.anneal("constant", [0.5], 300 * n, 150 * n) == ref[0][2][3]
    assert o.anneal("exponential", [2.0, 0.99999], 200 * n, BIG) == ref[0][3][3]
    assert o.anneal("constant", [1.0], 500 * n, 0) == ref[0][4][3]
    assert o.last_sweeps == 1
    assert o.anneal("constant", [1.0], 450 * n, 5) == ref[0][5][3]
    assert (o.memberships() == ref[1][3]).all()
    assert (o.last_accepted, o.last_sweeps) == (int(ref[3][0][3]), int(ref[3][1][3]))


def test_last_pass_steps_reports_the_kind_of_pass_and_a_new_partition_is_measured_afresh(monkeypatch):
    """bisbm_last_pass_steps: 1 / 2 / 4 / 8 steps per pass of the last launch -- what a pinned depth asks for (capped by what the
    block counts allow: 20 + 31 blocks: two or four; 40 + 33: two; 5 + 7: up to eight), 1 for the generic kernel (compat mode).
    The measured speeds of the depths belong to a partition: init / shuffle forget them, so the next launches try the depths
    again (the bench's equilibrated-start leg relies on it)."""
    mh = B.MetropolisHasting()
    na = nb = 6_000
    for ka, kb, pins in ((20, 31, {"2": 2, "4": 4, "8": 4}), (40, 33, {"2": 2, "4": 2}), (5, 7, {"2": 2, "4": 4, "8": 8})):
        rowptr, col = _random_graph(21, na, nb, 90_000, ka, kb)
        lab = O.contiguous_labels(na, nb, ka, kb)
        for pin, want in pins.items():
            monkeypatch.setenv("BISBM_PASS_DEPTH", pin)
            g = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, lab, n_chains=3, rng="philox", seed=5)
            g.shuffle_bisbm()
            assert g.last_pass_steps() == 0  # nothing launched yet
            mh.anneal(g, "constant", [1.0], 2 * (na + nb), BIG)
            assert g.last_pass_steps() == want, (ka, kb, pin)
        monkeypatch.setenv("BISBM_SINGLE_STEPS", "1")
        g = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, lab, n_chains=3, rng="philox", seed=5)
        g.shuffle_bisbm()
        mh.anneal(g, "constant", [1.0], na + nb, BIG)
        assert g.last_pass_steps() == 1
        monkeypatch.delenv("BISBM_SINGLE_STEPS")
    monkeypatch.delenv("BISBM_PASS_DEPTH")
    c = gpu_model(rowptr, col, na, nb, 5, 7, 1.0, lab, n_chains=1, rng="compat", seed=5, gen_seed=6)
    c.shuffle_bisbm()
    mh.anneal(c, "constant", [1.0], na + nb, BIG)
    assert c.last_pass_steps() == 1
    # free choice on 20 + 31 blocks: both depths are tried, again after every new partition
    rowptr, col = _random_graph(21, na, nb, 90_000, 20, 31)
    lab = O.contiguous_labels(na, nb, 20, 31)
    g = gpu_model(rowptr, col, na, nb, 20, 31, 1.0, lab, n_chains=3, rng="philox", seed=5)
    for start in (g.shuffle_bisbm, g.init_bisbm):
        start()
        tried = []
        for _ in range(4):  # (every depth twice before a measurement is trusted; the deepest first on a graph of this size)
            mh.anneal(g, "constant", [1.0], na + nb, BIG)
            tried.append(g.last_pass_steps())
        assert tried == [4, 4, 2, 2], tried


@pytest.mark.parametrize("roles", ["claims", "1", "2"])
def test_either_wave_can_step(roles, monkeypatch):
    """The production kernel settles at start which of a workgroup's two waves steps (per-SIMD claims); whichever it is
    -- wave 0 (BISBM_FIXED_ROLES=1), wave 1 (=2) or the claimed one -- the chains equal the oracle's."""
    if roles != "claims":
        monkeypatch.setenv("BISBM_FIXED_ROLES", roles)
    else:
        monkeypatch.delenv("BISBM_FIXED_ROLES", raising=False)
    name, na, nb, ne, ka, kb, eps, hubs, iso = next(c for c in CASES if c[0] == "hubs_isolated")
    rowptr, col = _random_graph(11, na, nb, ne, ka, kb, hubs, iso)
    labels = O.contiguous_labels(na, nb, ka, kb)
    n, chains, first = na + nb, 6, 7
    g = gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, n_chains=chains, rng="philox", seed=99, first_chain_id=first)
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    rates = mh.anneal(g, "constant", [1.0], 5 * n, BIG)
    rates2 = mh.anneal(g, "exponential", [2.0, 0.999], 5 * n, 3 * n)
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
        o.seed_philox(99, first + c)
        o.shuffle_bisbm()
        assert o.anneal("constant", [1.0], 5 * n, BIG) == rates[c]
        assert o.anneal("exponential", [2.0, 0.999], 5 * n, 3 * n) == rates2[c]
        assert_state_equal(g, o, c)


def test_early_stop_fires_in_production_kernel():
    """anneal() returns at the end of the sweep in which `steps_await` steps with T < 1 have passed without a new minimum
    (metropolis_hasting.cc:85-98).  The production kernel keeps that count as a difference of two counts; the run must
    stop in the same sweep as the oracle's literal bookkeeping, for several chains and schedules."""
    name, na, nb, ne, ka, kb, eps, hubs, iso = next(c for c in CASES if c[0] == "direct_tier")
    rowptr, col = _random_graph(11, na, nb, ne, ka, kb, hubs, iso)
    labels = O.contiguous_labels(na, nb, ka, kb)
    n, chains, first = na + nb, 5, 21
    g = gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, n_chains=chains, rng="philox", seed=4242, first_chain_id=first)
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    runs = [("exponential", [2.0, 0.9999], 40 * n, n // 2), ("abrupt_cool", [1.5 * n], 40 * n, n // 3),
            ("constant", [0.25], 40 * n, n // 4), ("linear", [1.2, 1.0 / (2 * n)], 40 * n, 0)]
    got = []
    for sched, kw, dur, await_ in runs:
        rates = mh.anneal(g, sched, kw, dur, await_)
        acc, sw = g.last_counts()
        got.append((rates.copy(), acc.copy(), sw.copy()))
    stopped_early = 0
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
        o.seed_philox(4242, first + c)
        o.shuffle_bisbm()
        for (sched, kw, dur, await_), (rates, acc, sw) in zip(runs, got):
            assert o.anneal(sched, kw, dur, await_) == rates[c], (sched, c)
            assert acc[c] == o.last_accepted and sw[c] == o.last_sweeps, (sched, c)
            stopped_early += int(o.last_sweeps < dur // n)
        assert_state_equal(g, o, c)
    assert stopped_early >= 3 * chains  # (the stop did fire: every run but possibly one per chain ended before its duration)


@pytest.mark.parametrize("case", ["direct_tier", "n_1000"])
def test_cooling_calls_run_as_table_slices(case, monkeypatch):
    """The production kernel holds no pow() / log(): the exponential and logarithmic schedules come from a host table
    (glibc, the reference's own values), and a call longer than the table runs as several launches of whole sweeps, each with
    its slice, the early-stop bookkeeping (metropolis_hasting.cc:75,85-98) carried over in the chain's scalars.  With the
    table capped at a few sweeps (BISBM_T_TABLE_CAP) every chain must equal its oracle run -- rates, counts, state -- whether
    the stop fires in the first slice, in a later one, or never, and equal the uncapped run."""
    if case == "n_1000":
        rowptr, col, na, nb = O.load_graph("n_1000")
        ka, kb, eps = 4, 6, 1.0
    else:
        name, na, nb, ne, ka, kb, eps, hubs, iso = next(c for c in CASES if c[0] == "direct_tier")
        rowptr, col = _random_graph(11, na, nb, ne, ka, kb, hubs, iso)
    labels = O.contiguous_labels(na, nb, ka, kb)
    n, chains, first = na + nb, 6, 5
    runs = [("exponential", [2.0, 0.99995], 37 * n + 11, 1 << 60), ("exponential", [2.0, 0.9999], 40 * n, n // 2),
            ("logarithmic", [1.0, 2.0], 23 * n, 1 << 60), ("logarithmic", [3.0, 2.0], 30 * n, 9 * n),
            ("exponential", [1e-3, 0.5], 12 * n, 1 << 60),   # underflows to T = 0 inside the call: the greedy tail
            ("exponential", [0.9, 0.99999], 31 * n, 0)]      # steps_await = 0 below T = 1: stops after the first sweep
    out = {}
    for cap in (str(3 * n + 17), str(1 << 22)):
        monkeypatch.setenv("BISBM_T_TABLE_CAP", cap)
        g = gpu_model(rowptr, col, na, nb, ka, kb, eps, labels, n_chains=chains, rng="philox", seed=777, first_chain_id=first)
        g.shuffle_bisbm()
        mh = B.MetropolisHasting()
        got = []
        for sched, kw, dur, await_ in runs:
            rates = mh.anneal(g, sched, kw, dur, await_).copy()
            acc, sw = g.last_counts()
            got.append((rates, acc.copy(), sw.copy()))
        out[cap] = (got, [g.get_memberships(c) for c in range(chains)], g.get_entropy().copy(), g)
    a, b = out[str(3 * n + 17)], out[str(1 << 22)]
    for (ra, aa, sa), (rb, ab, sb) in zip(a[0], b[0]):
        assert (ra == rb).all() and (aa == ab).all() and (sa == sb).all()
    assert all((x == y).all() for x, y in zip(a[1], b[1])) and (a[2] == b[2]).all()
    early = 0
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
        o.seed_philox(777, first + c)
        o.shuffle_bisbm()
        for (sched, kw, dur, await_), (rates, acc, sw) in zip(runs, a[0]):
            assert o.anneal(sched, kw, dur, await_) == rates[c], (sched, kw, c)
            assert (acc[c], sw[c]) == (o.last_accepted, o.last_sweeps), (sched, kw, c)
            early += int(3 < o.last_sweeps < dur // n)
        assert_state_equal(a[3], o, c)
    assert early >= 1  # (a stop fired in a later slice)


def test_config2_256_chains_philox():
    """BASELINE config 2: n_1000, Ka=4, Kb=6, 256 independent chains, constant T=1."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    labels = O.contiguous_labels(na, nb, 4, 6)
    chains, sweeps, first = 256, 4, 1000
    g = gpu_model(rowptr, col, na, nb, 4, 6, 1.0, labels, n_chains=chains, rng="philox", seed=2024, first_chain_id=first)
    g.shuffle_bisbm()
    rates = B.MetropolisHasting().anneal(g, B.constant_schedule, [1.0], sweeps * 1000, BIG)
    cum = g.get_entropy()
    ent = g.entropy()
    labs = []
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, 4, 6, 1.0, labels)
        o.seed_philox(2024, first + c)
        o.shuffle_bisbm()
        ro = o.anneal("constant", [1.0], sweeps * 1000, BIG)
        assert rates[c] == ro
        if c % 16 == 0:
            assert_state_equal(g, o, c)
            assert ent[c] == pytest.approx(o.entropy(), rel=1e-9)
        else:
            assert (g.get_memberships(c) == o.memberships()).all()
        assert sum_dS_close(cum[c], o)
        labs.append(o.memberships())
    # chains differ from each other
    assert len({tuple(l) for l in labs}) == chains
    # marginal histogram over the 256 chains
    g.marginals_reset()
    g.marginals_accumulate()
    want = B.distributed.numpy_marginals(np.array(labs), na, 4, 6)
    assert (g.marginals_get().astype(np.int64) == want).all()


def test_marginalize_driver_matches_oracle_samples():
    """README-style marginalisation (burn-in, samples every f sweeps, per-node histogram, MAP label) against
    the same schedule replayed chain by chain with the oracle."""
    rowptr, col, na, nb = O.load_graph("southernWomen")
    labels = O.contiguous_labels(na, nb, 5, 5)
    chains, burn, samples, freq = 16, 3, 5, 2
    g = gpu_model(rowptr, col, na, nb, 5, 5, 0.5, labels, n_chains=chains, rng="philox", seed=31)
    g.shuffle_bisbm()
    lab, counts = B.marginalize(g, burn, samples, freq)
    want = np.zeros((na + nb, 5), dtype=np.int64)
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, 5, 5, 0.5, labels)
        o.seed_philox(31, c)
        o.shuffle_bisbm()
        o.anneal("constant", [1.0], burn * 32, BIG)
        for _ in range(samples):
            o.anneal("constant", [1.0], freq * 32, BIG)
            want += B.distributed.numpy_marginals(o.memberships()[None, :], na, 5, 5)
    assert (counts == want).all() and counts.sum() == chains * samples * 32
    base = np.where(np.arange(32) >= na, 5, 0)
    assert (lab == want.argmax(axis=1) + base).all()
    # the same run into a caller-owned device tensor (the buffer RCCL reduces when chains are spread over ranks):
    # samples are ADDED to what the tensor holds, and the result is built from that tensor
    import torch
    g2 = gpu_model(rowptr, col, na, nb, 5, 5, 0.5, labels, n_chains=chains, rng="philox", seed=31)
    g2.shuffle_bisbm()
    dev = torch.full((na + nb, 5), 7, dtype=torch.int32, device=g2.counts_device())
    lab2, counts2 = B.marginalize(g2, burn, samples, freq, device_counts=dev)
    assert (counts2 == want + 7).all() and (dev.cpu().numpy() == want + 7).all()
    assert (lab2 == want.argmax(axis=1) + base).all()
    with pytest.raises(ValueError):
        B.marginalize(g2, 0, 1, 1, device_counts=dev.data_ptr())  # a raw pointer is refused


@pytest.mark.parametrize("ka,kb", [(170, 40), (200, 150)])
def test_marginal_histogram_with_many_blocks(ka, kb):
    """More than ~159 blocks of a type: a row of counters per thread no longer fits the LDS and the histogram is counted in
    HBM -- with byte labels (170 + 40) and in wide mode (200 + 150, two-byte labels)."""
    rowptr, col = cases.random_graph(17, 400, 300, 6000, ka, kb)
    na, nb = 400, 300
    g = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, O.contiguous_labels(na, nb, ka, kb), n_chains=5, rng="philox", seed=9)
    g.shuffle_bisbm()
    B.MetropolisHasting().anneal(g, "constant", [1.0], 2 * (na + nb), BIG)
    g.marginals_reset()
    g.marginals_accumulate(None)
    g.marginals_accumulate(None)
    want = 2 * B.distributed.numpy_marginals(np.stack([g.get_memberships(c) for c in range(5)]), na, ka, kb)
    assert (g.marginals_get() == want).all() and want.sum() == 2 * 5 * (na + nb)


# ------------------------------------------------------------------ agglomerative merges
def _merge_schedule(o, g, stages, mh, n):
    """mcmc_main.cc:379-396 / :425-444: one agg_merge per geospace stage, a greedy sweep (abrupt_cool, kwargs {0})
    between stages."""
    ka_s, kb_s = stages
    for i in range(len(ka_s) - 1):
        da, db = ka_s[i] - ka_s[i + 1], kb_s[i] - kb_s[i + 1]
        assert o.agg_merge(da, db, 10) == 0
        g.agg_merge(da, db, 10)
        assert (g.KA, g.KB) == (o.ka, o.kb) == (ka_s[i + 1], kb_s[i + 1])
        assert_state_equal(g, o)
        if i != len(ka_s) - 2:
            ro = o.anneal("abrupt_cool", [0.0], n, BIG)
            rg = mh.anneal(g, "abrupt_cool", [0.0], n, BIG)
            assert rg == ro
            assert_state_equal(g, o)


@pytest.mark.parametrize("mode", ["compat", "philox"])
def test_agg_merge_from_singletons_matches_oracle(mode):
    """--merge on southernWomen (mcmc_main.cc:349-399): every node its own block, merged down to 5 + 5."""
    rowptr, col, na, nb = O.load_graph("southernWomen")
    n = na + nb
    labels = np.arange(n, dtype=np.uint32)
    o = O.OracleModel(rowptr, col, na, nb, na, nb, 0.001, labels)
    if mode == "compat":
        o.seed_compat(42, 43)
        g = gpu_model(rowptr, col, na, nb, na, nb, 0.001, labels, rng="compat", seed=42, gen_seed=43)
    else:
        o.seed_philox(9, 2)
        g = gpu_model(rowptr, col, na, nb, na, nb, 0.001, labels, rng="philox", seed=9, first_chain_id=2)
    o.init_bisbm()
    g.init_bisbm()
    mh = B.MetropolisHasting()
    _merge_schedule(o, g, O.geospace(na, 5, nb, 5, 1.01), mh, n)
    ro = o.anneal("abrupt_cool", [50.0], 20 * n, BIG)
    rg = mh.anneal(g, "abrupt_cool", [50.0], 20 * n, BIG)
    assert rg == ro
    assert_state_equal(g, o)
    assert g.entropy()[0] == pytest.approx(o.entropy(), rel=1e-12)


@pytest.mark.parametrize("mode", ["compat", "philox"])
def test_agg_merge_down_from_initial_partition(mode):
    """mcmc_main.cc:402-450: an initial partition with more blocks than asked for (12 + 15 on n_1000) merged down
    to 4 + 6 in geometric stages; also the one-shot merge over both types (blockmodel.cc:208-271)."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    n = na + nb
    labels = O.contiguous_labels(na, nb, 12, 15)
    o = O.OracleModel(rowptr, col, na, nb, 12, 15, 1.0, labels)
    if mode == "compat":
        o.seed_compat(5, 6)
        g = gpu_model(rowptr, col, na, nb, 12, 15, 1.0, labels, rng="compat", seed=5, gen_seed=6)
    else:
        o.seed_philox(31, 0)
        g = gpu_model(rowptr, col, na, nb, 12, 15, 1.0, labels, rng="philox", seed=31)
    o.shuffle_bisbm()
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    _merge_schedule(o, g, O.geospace(12, 6, 15, 8, 1.2), mh, n)
    assert o.agg_merge_total(4, 10) == 0   # --nature style: 4 merges wherever they are cheapest
    g.agg_merge(4, None, 10)
    assert (g.KA, g.KB) == (o.ka, o.kb) and g.K == 10
    assert_state_equal(g, o)
    ro = o.anneal("constant", [1.0], 3 * n, BIG)
    rg = mh.anneal(g, "constant", [1.0], 3 * n, BIG)
    assert rg == ro
    assert_state_equal(g, o)
    with pytest.raises(B.BisbmError):      # the one-argument overload has no split branch (blockmodel.cc:208-271)
        g.agg_merge(-1, None, 10)


def test_agg_merge_many_chains_philox():
    """64 chains merged at once; every chain equals its own oracle run."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    labels = O.contiguous_labels(na, nb, 8, 9)
    chains = 64
    g = gpu_model(rowptr, col, na, nb, 8, 9, 1.0, labels, n_chains=chains, rng="philox", seed=77, first_chain_id=10)
    g.shuffle_bisbm()
    g.agg_merge(4, 3, 10)
    assert (g.KA, g.KB) == (4, 6)
    for c in range(0, chains, 7):
        o = O.OracleModel(rowptr, col, na, nb, 8, 9, 1.0, labels)
        o.seed_philox(77, 10 + c)
        o.shuffle_bisbm()
        assert o.agg_merge(4, 3, 10) == 0
        assert_state_equal(g, o, c)


@pytest.mark.parametrize("mode", ["compat", "philox"])
def test_agg_split_matches_oracle(mode):
    """agg_split (blockmodel.cc:505-565, intended rank-within-block semantics) through negative diffs of
    agg_merge (:110-117): one split per unit, type a first; then sweeps on the wider partition; then a call that
    splits one type and merges the other."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    n = na + nb
    labels = O.contiguous_labels(na, nb, 3, 4)
    o = O.OracleModel(rowptr, col, na, nb, 3, 4, 1.0, labels)
    if mode == "compat":
        o.seed_compat(21, 22)
        g = gpu_model(rowptr, col, na, nb, 3, 4, 1.0, labels, rng="compat", seed=21, gen_seed=22)
    else:
        o.seed_philox(21, 6)
        g = gpu_model(rowptr, col, na, nb, 3, 4, 1.0, labels, rng="philox", seed=21, first_chain_id=6)
    o.shuffle_bisbm()
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    assert mh.anneal(g, "constant", [1.0], 2 * n, BIG) == o.anneal("constant", [1.0], 2 * n, BIG)
    assert o.agg_merge(-1, -2, 7) == 0
    g.agg_merge(-1, -2, 7)
    assert (g.KA, g.KB) == (o.ka, o.kb) == (4, 6)
    assert_state_equal(g, o)
    assert g.entropy()[0] == pytest.approx(o.entropy(), rel=1e-12)
    assert mh.anneal(g, "constant", [1.0], 3 * n, BIG) == o.anneal("constant", [1.0], 3 * n, BIG)
    assert_state_equal(g, o)
    assert o.agg_merge(-1, 2, 10) == 0  # one more type-a block, two type-b blocks fewer
    g.agg_merge(-1, 2, 10)
    assert (g.KA, g.KB) == (o.ka, o.kb) == (5, 4)
    assert_state_equal(g, o)
    assert mh.anneal(g, "abrupt_cool", [0.0], n, BIG) == o.anneal("abrupt_cool", [0.0], n, BIG)
    assert_state_equal(g, o)
    # marginals on the wider partition (the histogram has max(KA, KB) columns)
    g.marginals_reset()
    g.marginals_accumulate()
    counts = g.marginals_get()
    assert counts.shape == (n, 5) and (counts.sum(axis=1) == 1).all()


def test_agg_split_many_chains_philox():
    """48 chains split at once (type b, then type a); sampled chains equal their own oracle runs."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    labels = O.contiguous_labels(na, nb, 4, 5)
    chains = 48
    g = gpu_model(rowptr, col, na, nb, 4, 5, 1.0, labels, n_chains=chains, rng="philox", seed=8, first_chain_id=3)
    g.shuffle_bisbm()
    g.agg_merge(0, -1, 12)
    g.agg_merge(-1, 0, 12)
    assert (g.KA, g.KB) == (5, 6)
    for c in range(0, chains, 11):
        o = O.OracleModel(rowptr, col, na, nb, 4, 5, 1.0, labels)
        o.seed_philox(8, 3 + c)
        o.shuffle_bisbm()
        assert o.agg_split(1, 12) == 0 and o.agg_split(0, 12) == 0
        assert_state_equal(g, o, c)
    # refusals: 256 blocks is the label format's limit; a partition of single nodes cannot be split
    rp, cl, a2, b2 = O.load_graph("southernWomen")
    t = gpu_model(rp, cl, a2, b2, a2, b2, 0.001, np.arange(a2 + b2, dtype=np.uint32), rng="philox", seed=1)
    t.init_bisbm()
    with pytest.raises(B.BisbmError):
        t.agg_merge(-1, 0, 5)


@pytest.mark.parametrize("mode", ["philox", "compat"])
def test_chains_may_end_with_different_block_counts(mode):
    """agg_merge(engine, diff, nm) lets every run end with its own (Ka,Kb) (blockmodel.cc:208-271).  The handle then keeps
    its chains grouped by shape: every chain stays equal to its own oracle run through a sweep, a second one-argument
    merge (groups split further), a two-argument merge and another sweep; per-chain shapes, states, rates, sum dS and
    description lengths are served, the calls that need one common shape say so."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    n = na + nb
    labels = O.contiguous_labels(na, nb, 8, 9)
    chains = 12
    g = gpu_model(rowptr, col, na, nb, 8, 9, 1.0, labels, n_chains=chains, rng=mode, seed=123, gen_seed=77)
    g.shuffle_bisbm()
    mh = B.MetropolisHasting()
    mh.anneal(g, "constant", [1.0], 3 * n, BIG)  # (right after a shuffle every chain merges type b only)
    os_ = []
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, 8, 9, 1.0, labels)
        if mode == "philox":
            o.seed_philox(123, c)
        else:
            o.seed_compat(123 + c, 77 + c)
        o.shuffle_bisbm()
        o.anneal("constant", [1.0], 3 * n, BIG)
        os_.append(o)

    def check():
        for c, o in enumerate(os_):
            assert g.ka_kb(c) == (o.ka, o.kb)
            assert_state_equal(g, o, c)
        ent, cum = g.entropy(), g.get_entropy()
        for c, o in enumerate(os_):
            assert abs(ent[c] - o.entropy()) <= 1e-9 * abs(o.entropy())
            assert sum_dS_close(cum[c], o) or abs(cum[c] - o.get_entropy()) <= 1e-9

    g.agg_merge(5, None, 10)
    for o in os_:
        assert o.agg_merge_total(5, 10) == 0
    assert len({(o.ka, o.kb) for o in os_}) > 1 and g.mixed_shapes  # the premise: these chains do not agree
    check()
    rg = mh.anneal(g, "constant", [1.0], 2 * n, BIG)
    for c, o in enumerate(os_):
        assert rg[c] == o.anneal("constant", [1.0], 2 * n, BIG)
    check()
    g.agg_merge(3, None, 10)  # groups split further
    for o in os_:
        assert o.agg_merge_total(3, 10) == 0
    check()
    g.agg_merge(1, 1, 10)  # the two-argument overload: the same change in every group
    for o in os_:
        assert o.agg_merge(1, 1, 10) == 0
    rg = mh.anneal(g, "abrupt_cool", [0.0], n, BIG)
    for c, o in enumerate(os_):
        assert rg[c] == o.anneal("abrupt_cool", [0.0], n, BIG)
    check()
    # one common shape is gone: the calls that need it say so
    ka, kb = C.c_uint32(), C.c_uint32()
    assert g._L.bisbm_get_ka_kb(g._h, C.byref(ka), C.byref(kb)) == B.BISBM_ERR_STATE
    with pytest.raises(B.BisbmError) as e:
        g.marginals_accumulate(None)
    assert e.value.code == B.BISBM_ERR_STATE
    # labels can still be set per chain (they must name blocks of that chain's shape), and the state rebuilt
    g.set_memberships(os_[3].memberships(), chain=3)
    g.init_bisbm()
    assert_state_equal(g, os_[3], 3)


def test_cli_nature_with_several_chains():
    """`mcmc --merge --nature --chains 3` (mcmc_main.cc:354-377,398-404): every chain may end with its own (Ka,Kb), so
    the shell runs them in handles of their own and prints "KA KB labels" of the chain with the lowest description
    length -- replayed with three oracle runs."""
    import math
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    r = subprocess.run([cli, "-e", os.path.join(O.GOLDEN, "southernWomen.edgelist"), "-y", "18", "14", "-n", "18", "14", "-z", "1", "1",
                        "--merge", "--nature", "--chains", "3", "-t", "640", "-x", "100000", "-c", "abrupt_cool", "-a", "320",
                        "-E", "0.001", "-d", "42", "--gen_seed", "43"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rowptr, col, na, nb = O.load_graph("southernWomen")
    n = na + nb
    best = None
    for c in range(3):
        o = O.OracleModel(rowptr, col, na, nb, na, nb, 0.001, np.arange(n, dtype=np.uint32))
        o.seed_compat(42 + c, 43 + c)
        o.init_bisbm()
        ceiling = math.ceil(math.sqrt(2.0 * o.num_edges) / 2)
        tka, tkb = na, nb
        while tka >= ceiling and tkb >= ceiling:
            assert o.agg_merge_total(math.ceil((tka + tkb) * (1.01 - 1) / 1.01), 10) == 0
            tka, tkb = o.ka, o.kb
            o.anneal("abrupt_cool", [0.0], n, 100000)
        o.anneal("abrupt_cool", [320.0], 640, 100000)
        dl = o.entropy()
        if best is None or dl < best[0]:
            best = (dl, o.ka, o.kb, o.memberships().copy(), c)
    assert r.stdout == "%d %d " % (best[1], best[2]) + " ".join(map(str, best[3])) + " \n"
    assert "printing chain %d" % best[4] in r.stderr and "(Ka, Kb) = (%d, %d) " % (best[1], best[2]) in r.stderr


def test_cli_split_matches_oracle_replay():
    """`mcmc` with -z larger than the initial partition (mcmc_main.cc:419-451, else branch: agg_merge(diff_a, diff_b,
    100) with negative diffs = agg_split, then the final anneal) against the same driver replayed with the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    r = subprocess.run([cli, "-e", os.path.join(O.GOLDEN, "southernWomen.edgelist"), "-y", "18", "14", "-n", "9", "9", "7", "7",
                        "-z", "3", "4", "-t", "3200", "-x", "100000", "-c", "abrupt_cool", "-a", "320", "-E", "0.001",
                        "-d", "42", "--gen_seed", "43"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rowptr, col, na, nb = O.load_graph("southernWomen")
    o = O.OracleModel(rowptr, col, na, nb, 2, 2, 0.001, O.labels_from_sizes([9, 9, 7, 7]))
    o.seed_compat(42, 43)
    o.init_bisbm()
    assert o.agg_merge(2 - 3, 2 - 4, 100) == 0
    o.anneal("abrupt_cool", [320.0], 3200, 100000)
    assert r.stdout == " ".join(map(str, o.memberships())) + " \n"
    assert "(Ka, Kb) = (3, 4) " in r.stderr


def test_running_sum_from_the_description_length_equals_the_sum_of_the_steps(monkeypatch):
    """A production launch that cannot stop early does not add up its accepted dS step by step: bisbm_anneal advances the chain's
    sum by the change of the block-state part of the description length over the call.  BISBM_KEEP_SUM=1 keeps the step-by-step
    sum (the code path of the early-stop bookkeeping).  Same chains both ways: equal labels and rates, sums equal to a few ulps
    of the description length S -- and each within the tolerance of the oracle's step-by-step sum."""
    mh = B.MetropolisHasting()
    rowptr, col = _random_graph(31, 6000, 6000, 150_000, 24, 31)
    lab = O.contiguous_labels(6000, 6000, 24, 31)
    n = 12_000
    out = {}
    for keep in ("0", "1"):
        monkeypatch.setenv("BISBM_KEEP_SUM", keep)
        g = gpu_model(rowptr, col, 6000, 6000, 24, 31, 1.0, lab, n_chains=5, rng="philox", seed=321)
        g.shuffle_bisbm()
        rates = [mh.anneal(g, "constant", [1.0], 3 * n, BIG).copy(), mh.anneal(g, "exponential", [1.5, 0.9999], 2 * n, BIG).copy(),
                 mh.anneal(g, "exponential", [0.9, 0.9999], 2 * n, n // 2).copy(),  # (early stop in reach: step-by-step either way)
                 mh.anneal(g, "constant", [2.0], n + 17, BIG).copy()]
        out[keep] = (rates, [g.get_memberships(c) for c in range(5)], g.get_entropy().copy(), g.entropy().copy())
    monkeypatch.delenv("BISBM_KEEP_SUM")
    a, b = out["0"], out["1"]
    assert all((x == y).all() for x, y in zip(a[0], b[0])) and all((x == y).all() for x, y in zip(a[1], b[1]))
    assert (a[3] == b[3]).all()
    assert (np.abs(a[2] - b[2]) <= 1e-12 * np.abs(a[3])).all(), (a[2], b[2])
    o = O.OracleModel(rowptr, col, 6000, 6000, 24, 31, 1.0, lab)
    o.seed_philox(321, 2)
    o.shuffle_bisbm()
    for (s, kw, dur, aw), r in zip([("constant", [1.0], 3 * n, BIG), ("exponential", [1.5, 0.9999], 2 * n, BIG),
                                    ("exponential", [0.9, 0.9999], 2 * n, n // 2), ("constant", [2.0], n + 17, BIG)], a[0]):
        assert o.anneal(s, kw, dur, aw) == r[2]
    assert sum_dS_close(a[2][2], o) and sum_dS_close(b[2][2], o)
    assert abs(b[2][2] - o.get_entropy()) <= 1e-9 * abs(o.get_entropy())  # (step by step: the tolerance of the sum itself)


def test_anneal_splits_compose_on_device():
    rowptr, col, na, nb = O.load_graph("n_1000")
    labels = O.contiguous_labels(na, nb, 4, 6)
    mh = B.MetropolisHasting()
    for mode, kw in (("compat", dict(seed=9, gen_seed=10)), ("philox", dict(seed=9))):
        a = gpu_model(rowptr, col, na, nb, 4, 6, 1.0, labels, rng=mode, **kw)
        b = gpu_model(rowptr, col, na, nb, 4, 6, 1.0, labels, rng=mode, **kw)
        a.shuffle_bisbm()
        b.shuffle_bisbm()
        mh.anneal(a, "constant", [1.0], 5000, BIG)
        for _ in range(5):
            mh.anneal(b, "constant", [1.0], 1000, BIG)
        assert (a.get_memberships() == b.get_memberships()).all()
        if mode == "compat":
            assert a.get_entropy()[0] == b.get_entropy()[0]
        else:  # (production kernel without the early-stop bookkeeping: the sum advances per CALL by the change of the description
               # length, so one call and five calls agree to rounding, not to the bit)
            assert a.get_entropy()[0] == pytest.approx(b.get_entropy()[0], rel=1e-12, abs=1e-9)


def test_device_log_q_matches_oracle():
    rowptr, col, na, nb = O.load_graph("n_1000")
    g = gpu_model(rowptr, col, na, nb, 4, 6, 1.0, O.contiguous_labels(na, nb, 4, 6))
    L = O.lib()
    rng = np.random.default_rng(3)
    # table branch (n <= 10000), approximation branch (k >= n^(1/4)) at any n, and the small-k branch
    # (lbinom through the lgamma table) for n inside this model's table (2E+2 = 30002 entries)
    n_big = rng.integers(10001, 2_000_000, 3000)
    k_big = np.maximum(rng.integers(1, 100_000, 3000), np.ceil(n_big ** 0.25).astype(np.int64) + 1)
    n = np.concatenate([rng.integers(1, 10001, 3000), n_big,
                        [0, -5, 7, 10000, 10001, 312500, 20000, 25000, 29000, 160000]]).astype(np.int32)
    k = np.concatenate([np.minimum(rng.integers(1, 501, 3000), 500), k_big,
                        [3, 3, 0, 500, 500, 15625, 5, 11, 2, 21]]).astype(np.int32)
    got = g.debug_log_q(n, k)
    want = np.array([L.orc_log_q(int(a), int(b)) for a, b in zip(n, k)])
    table = n < 10001
    assert (got[table] == want[table]).all()  # host-built table: same bits
    assert np.allclose(got[~table], want[~table], rtol=1e-12, atol=0)  # device libm vs glibc
    # Philox-mode evaluation (production definition): get_v taken to convergence for u = k/sqrt(n) >= 2.5, evaluated by
    # closed forms (u > 18: log_q_closed, 13..18: log_q_closed2, 8..13: log_q_mid, 2.5..8: log_q_low); the literal code below 2.5.
    n_mid = rng.integers(10001, 20_000_000, 6000)
    k_mid = np.maximum(1, np.round(rng.uniform(1.5, 26.0, 6000) * np.sqrt(n_mid))).astype(np.int64)
    n = np.concatenate([n, n_mid]).astype(np.int32)
    k = np.concatenate([k, k_mid]).astype(np.int32)
    table = n < 10001
    fast = g.debug_log_q(n, k, fast=True)
    # (1) the device equals the CPU restatement of that definition to a few ulp
    want_phx = np.array([L.orc_log_q_philox(int(a), int(b)) for a, b in zip(n, k)])
    assert (fast[table] == want_phx[table]).all()
    assert np.allclose(fast[~table], want_phx[~table], rtol=2e-15, atol=0)
    # (2) and differs from the reference's literal evaluation only by what the literal's |dv| <= 1e-8 stop leaves
    want = np.array([L.orc_log_q(int(a), int(b)) for a, b in zip(n, k)])
    nt, kt = n[~table].astype(np.float64), np.minimum(k[~table], n[~table]).astype(np.float64)
    u2 = kt * kt / nt
    rel = np.abs(fast[~table] - want[~table]) / np.abs(want[~table])
    tol = np.select([u2 >= 169, u2 >= 100, u2 >= 64, u2 >= 36, u2 >= 16, u2 >= 6.25], [2e-15, 3e-14, 2e-12, 2e-11, 2e-10, 8e-10], 2e-15)
    assert (rel <= tol).all(), (rel / tol).max()
    assert (u2 > 576).sum() > 300 and ((u2 >= 64) & (u2 <= 576)).sum() > 3000 and ((u2 >= 6.25) & (u2 < 64)).sum() > 1000 \
        and (u2 < 6.25).sum() > 200


def test_error_paths():
    rowptr, col, na, nb = O.load_graph("southernWomen")
    labels = O.labels_from_sizes([4, 4, 4, 3, 3, 3, 3, 3, 3, 2])
    with pytest.raises(B.BisbmError) as e:
        gpu_model(rowptr, col, na + 1, nb - 1, 5, 5, 1.0, labels)  # node 18 becomes type a: a-a edges
    assert e.value.code == B.BISBM_ERR_NOT_BIPARTITE
    with pytest.raises(B.BisbmError) as e:
        gpu_model(rowptr, col, na, nb, 200, 100, 1.0, labels)  # more blocks than nodes
    assert e.value.code == B.BISBM_ERR_INVALID_ARG
    with pytest.raises(B.BisbmError) as e:  # more than 65535 blocks: labels are at most two bytes
        big_n = 70000
        gpu_model(np.zeros(big_n + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32), 40000, 30000, 40000, 30000, 1.0,
                  np.arange(big_n, dtype=np.uint32))
    assert e.value.code == B.BISBM_ERR_UNSUPPORTED
    bad = labels.copy()
    bad[0] = 7  # a type-b block for a type-a node
    with pytest.raises(B.BisbmError) as e:
        gpu_model(rowptr, col, na, nb, 5, 5, 1.0, bad)
    assert e.value.code == B.BISBM_ERR_INVALID_ARG
    m = gpu_model(rowptr, col, na, nb, 5, 5, 1.0, labels)
    with pytest.raises(B.BisbmError) as e:
        B.MetropolisHasting().anneal(m, "constant", [1.0], 32, BIG)  # no init_bisbm / shuffle_bisbm yet
    assert e.value.code == B.BISBM_ERR_STATE
    with pytest.raises(B.BisbmError) as e:
        gpu_model(rowptr, col, na, nb, 5, 5, 1.0, labels, device=99)
    assert e.value.code == B.BISBM_ERR_NO_DEVICE


def test_io_round_trip(tmp_path):
    """Host I/O of the product library against the oracle's restatement (itself checked against the
    reference's graph_utilities.cc in the CPU suite)."""
    p = tmp_path / "q.el"
    p.write_text("0\t5\n1 6\n\n2   7\r\nabc def\n3\n4 8 junk\n")
    a, b = B.load_edge_list(str(p))
    oa, ob = O.load_edge_list(str(p))
    assert (a == oa).all() and (b == ob).all()
    r1, c1 = B.edge_to_adj((a, b), 9)
    r2, c2 = O.edge_to_csr(oa, ob, 9)
    assert (r1 == r2).all() and (c1 == c2).all()
    assert B.output_vec([3, 0, 12], stream=open(os.devnull, "w")) == "3 0 12 \n"


def test_cli_golden(golden):
    """The re-hosted `mcmc` command line prints the reference's recorded labels for config 1
    (engine seed 42, gen seed 43): stdout is the label line of output_vec, byte for byte."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    g = golden["compat_rng"]["scenario1_config1"]
    r = subprocess.run([cli, "-e", os.path.join(O.GOLDEN, "southernWomen.edgelist"), "-n", "4", "4", "3", "4", "3", "3",
                        "3", "3", "3", "2", "-y", "18", "14", "-z", "5", "5", "-t", "32000", "-x", "100", "-c",
                        "exponential", "-a", "10", "0.1", "-E", "0.001", "--randomize", "-d", "42", "--gen_seed", "43"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout == " ".join(map(str, g["labels"])) + " \n"
    assert "acceptance ratio 0.248264" in r.stderr and "(Ka, Kb) = (5, 5) " in r.stderr and "entropy: 221.095" in r.stderr


def test_cli_merge_matches_oracle_replay():
    """`mcmc --merge` (mcmc_main.cc:349-399) on southernWomen against the same driver replayed with the oracle:
    one block per node, staged merges down to 5 + 5 with greedy sweeps in between, final abrupt_cool anneal."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    r = subprocess.run([cli, "-e", os.path.join(O.GOLDEN, "southernWomen.edgelist"), "-y", "18", "14", "-n", "18", "14", "-z", "5", "5",
                        "--merge", "-t", "3200", "-x", "100000", "-c", "abrupt_cool", "-a", "320", "-E", "0.001",
                        "-d", "42", "--gen_seed", "43"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rowptr, col, na, nb = O.load_graph("southernWomen")
    n = na + nb
    o = O.OracleModel(rowptr, col, na, nb, na, nb, 0.001, np.arange(n, dtype=np.uint32))
    o.seed_compat(42, 43)
    o.init_bisbm()
    ka_s, kb_s = O.geospace(na, 5, nb, 5, 1.01)
    for i in range(len(ka_s) - 1):
        assert o.agg_merge(ka_s[i] - ka_s[i + 1], kb_s[i] - kb_s[i + 1], 10) == 0
        if i != len(ka_s) - 2:
            o.anneal("abrupt_cool", [0.0], n, 100000)
    o.anneal("abrupt_cool", [320.0], 3200, 100000)
    assert r.stdout == " ".join(map(str, o.memberships())) + " \n"
    assert "(Ka, Kb) = (5, 5) " in r.stderr


def test_cli_merge_from_singletons_on_1000_nodes():
    """`mcmc --merge` on the n_1000 data set: the run starts at one block per node (KA + KB = 1000: the library's wide
    mode, two-byte labels and m in HBM), 463 merge stages with greedy sweeps in between bring it to 4 + 6 (byte labels and
    the ordinary kernels from 256 blocks down), then the final anneal -- against the same driver replayed with the oracle."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    el = os.path.join(O.GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist")
    r = subprocess.run([cli, "-e", el, "-y", "500", "500", "-n", "500", "500", "-z", "4", "6", "--merge", "-t", "10000", "-x", "100000",
                        "-c", "abrupt_cool", "-a", "100", "-E", "1", "-d", "42", "--gen_seed", "43"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rowptr, col = B.load_graph(el, 1000)
    na = nb = 500
    n = na + nb
    o = O.OracleModel(rowptr, col, na, nb, na, nb, 1.0, np.arange(n, dtype=np.uint32))
    o.seed_compat(42, 43)
    o.init_bisbm()
    ka_s, kb_s = O.geospace(na, 4, nb, 6, 1.01)
    for i in range(len(ka_s) - 1):
        assert o.agg_merge(ka_s[i] - ka_s[i + 1], kb_s[i] - kb_s[i + 1], 10) == 0
        if i != len(ka_s) - 2:
            o.anneal("abrupt_cool", [0.0], n, 100000)
    o.anneal("abrupt_cool", [100.0], 10000, 100000)
    assert r.stdout == " ".join(map(str, o.memberships())) + " \n"
    assert "(Ka, Kb) = (4, 6) " in r.stderr


def test_cli_merge_nature_on_1000_nodes():
    """`mcmc --merge --nature` on n_1000 (mcmc_main.cc:354-377): one-argument agg_merge stages from 1000 blocks (wide mode)
    down to fewer than sqrt(2E)/2 of a type, greedy sweeps in between -- against the oracle's replay."""
    import math
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    el = os.path.join(O.GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist")
    r = subprocess.run([cli, "-e", el, "-y", "500", "500", "-n", "500", "500", "-z", "1", "1", "--merge", "--nature", "-t", "2000",
                        "-x", "100000", "-c", "abrupt_cool", "-a", "100", "-E", "1", "-d", "11", "--gen_seed", "12"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rowptr, col = B.load_graph(el, 1000)
    na = nb = 500
    n = na + nb
    o = O.OracleModel(rowptr, col, na, nb, na, nb, 1.0, np.arange(n, dtype=np.uint32))
    o.seed_compat(11, 12)
    o.init_bisbm()
    ceiling = math.ceil(math.sqrt(2.0 * o.num_edges) / 2)
    tka, tkb = na, nb
    while tka >= ceiling and tkb >= ceiling:
        assert o.agg_merge_total(math.ceil((tka + tkb) * (1.01 - 1) / 1.01), 10) == 0
        tka, tkb = o.ka, o.kb
        o.anneal("abrupt_cool", [0.0], n, 100000)
    o.anneal("abrupt_cool", [100.0], 2000, 100000)
    assert r.stdout == "%d %d " % (o.ka, o.kb) + " ".join(map(str, o.memberships())) + " \n"


def test_wide_mode_merges_down_into_byte_labels():
    """API level, both RNG modes, several chains: 150 + 150 blocks (wide) split to 151 + 151 (agg_split while wide, round 3),
    merged to 141 + 141 (still wide), swept, merged to 101 + 101 (byte labels from here), swept -- state equal to the oracle's after
    every call; the marginal histogram is served there too."""
    rowptr, col = cases.random_graph(91, 300, 300, 5000, 150, 150)
    na = nb = 300
    n = na + nb
    labels = O.contiguous_labels(na, nb, 150, 150)
    mh = B.MetropolisHasting()
    for mode in ("compat", "philox"):
        chains = 3
        g = gpu_model(rowptr, col, na, nb, 150, 150, 1.0, labels, n_chains=chains, rng=mode, seed=5, gen_seed=6)
        g.shuffle_bisbm()
        os_ = []
        for c in range(chains):
            o = O.OracleModel(rowptr, col, na, nb, 150, 150, 1.0, labels)
            if mode == "compat":
                o.seed_compat(5 + c, 6 + c)
            else:
                o.seed_philox(5, c)
            o.shuffle_bisbm()
            os_.append(o)
        g.agg_merge(-1, -1, 5)  # a split of each type above 256 blocks (blockmodel.cc:110-117 -> agg_split :505-565)
        for c, o in enumerate(os_):
            assert o.agg_merge(-1, -1, 5) == 0
            assert_state_equal(g, o, c)
        assert (g.KA, g.KB) == (151, 151)
        g.marginals_reset()  # (the histogram is counted in HBM when a row of counters per thread no longer fits the LDS)
        g.marginals_accumulate(None)
        want = B.distributed.numpy_marginals(np.stack([o.memberships() for o in os_]), na, 151, 151)
        assert (g.marginals_get() == want).all()
        for (da, db), sched in (((10, 10), ("constant", [1.0])), ((40, 40), ("abrupt_cool", [0.0]))):
            g.agg_merge(da, db, 10)
            for o in os_:
                assert o.agg_merge(da, db, 10) == 0
            rg = mh.anneal(g, sched[0], sched[1], 2 * n, BIG)
            for c, o in enumerate(os_):
                ro = o.anneal(sched[0], sched[1], 2 * n, BIG)
                assert rg[c] == ro
                assert_state_equal(g, o, c)
        assert (g.KA, g.KB) == (101, 101)
        ent = g.entropy()
        for c, o in enumerate(os_):
            assert abs(ent[c] - o.entropy()) <= 1e-9 * abs(o.entropy())


def test_cli_resume_round_trip(tmp_path):
    """The reference's de-facto checkpoint: print labels, feed them back with --membership_path
    (mcmc_main.cc:247-278: randomize forced off, KA = max type-a label + 1, KB = max label - max type-a label, -n and
    -z not needed).  A zero-step continuation must print the same labels and the same description length, i.e.
    init_bisbm() rebuilt the very state the first run ended in; the edge list comes through the binary CSR cache the
    second and third time."""
    import re
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    el = str(tmp_path / "n1000.edgelist")
    shutil.copy(os.path.join(O.GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist"), el)
    first = subprocess.run([cli, "-e", el, "-y", "500", "500", "-n", "125", "125", "125", "125", "84", "84", "83", "83", "83", "83",
                            "-z", "4", "6", "-t", "20000", "-x", "100000", "-c", "constant", "-a", "1", "-E", "1", "--randomize",
                            "-d", "7", "--gen_seed", "8"], capture_output=True, text=True)
    assert first.returncode == 0, first.stderr
    labels = first.stdout.split()
    assert len(labels) == 1000
    mpath = tmp_path / "resume.txt"
    mpath.write_text("\n".join(labels) + "\n")
    ent = re.search(r"entropy: (\S+)", first.stderr).group(1)

    def resume(*extra):
        r = subprocess.run([cli, "-e", el, "-y", "500", "500", "--membership_path", str(mpath), "-t", "0", "-c", "constant",
                            "-a", "1", "-E", "1", "-d", "99", *extra], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        return r
    again = resume()
    assert again.stdout == first.stdout
    assert "(Ka, Kb) = (4, 6) " in again.stderr and "entropy: %s" % ent in again.stderr
    assert " ---- read membership from file! ---- " in again.stderr
    cached = resume("--csr_cache")           # writes <edge list>.bisbm_csr
    assert os.path.exists(el + ".bisbm_csr") and cached.stdout == first.stdout
    cached2 = resume("--csr_cache")          # reads it
    assert cached2.stdout == first.stdout and "entropy: %s" % ent in cached2.stderr
    # and through the library: the state init_bisbm() builds from the printed labels is the state of the oracle run
    rowptr, col = B.load_graph(el, 1000, cache=True)
    assert B.load_graph.last_cache_hit
    o = O.OracleModel(rowptr, col, 500, 500, 4, 6, 1.0, O.labels_from_sizes([125] * 4 + [84, 84, 83, 83, 83, 83]))
    o.seed_compat(7, 8)
    o.shuffle_bisbm()
    o.anneal("constant", [1.0], 20000, 100000)
    assert [int(x) for x in labels] == list(o.memberships())
    g = gpu_model(rowptr, col, 500, 500, 4, 6, 1.0, np.array(labels, dtype=np.uint32), rng="compat", seed=1)
    g.init_bisbm()
    assert_state_equal(g, o)
    # a continuation from the file is a fresh process: new engines, visit list back at 0..N-1 (blockmodel.cc:41)
    cont = subprocess.run([cli, "-e", el, "-y", "500", "500", "--membership_path", str(mpath), "-t", "5000", "-x", "100000",
                           "-c", "constant", "-a", "1", "-E", "1", "-d", "31", "--gen_seed", "32"], capture_output=True, text=True)
    assert cont.returncode == 0, cont.stderr
    o2 = O.OracleModel(rowptr, col, 500, 500, 4, 6, 1.0, np.array(labels, dtype=np.uint32))
    o2.seed_compat(31, 32)
    o2.init_bisbm()
    o2.anneal("constant", [1.0], 5000, 100000)
    assert cont.stdout == " ".join(map(str, o2.memberships())) + " \n"


def test_cli_reorder_runs_the_renumbered_graph_and_prints_the_callers_numbering():
    """`mcmc --reorder`: the engine runs on the graph renumbered by bisbm_io_locality_order, with the initial labels
    carried along; stdout is in the caller's numbering.  Replayed with the oracle on the renumbered graph."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    el = os.path.join(O.GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist")
    sizes = [125] * 4 + [84, 84, 83, 83, 83, 83]
    r = subprocess.run([cli, "-e", el, "-y", "500", "500", "-n", *map(str, sizes), "-z", "4", "6", "-t", "10000", "-x", "100000",
                        "-c", "constant", "-a", "1", "-E", "1", "-d", "3", "--gen_seed", "4", "--reorder"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    rowptr, col = B.load_graph(el, 1000)
    lo = B.locality_order(rowptr, col, 500, 500)
    rp2, cl2 = lo.apply(rowptr, col)
    o = O.OracleModel(rp2, cl2, 500, 500, 4, 6, 1.0, lo.to_new(O.labels_from_sizes(sizes)))
    o.seed_compat(3, 4)
    o.init_bisbm()
    o.anneal("constant", [1.0], 10000, 100000)
    assert r.stdout == " ".join(map(str, lo.to_old(o.memberships()))) + " \n"
    plain = subprocess.run([cli, "-e", el, "-y", "500", "500", "-n", *map(str, sizes), "-z", "4", "6", "-t", "10000", "-x", "100000",
                            "-c", "constant", "-a", "1", "-E", "1", "-d", "3", "--gen_seed", "4"], capture_output=True, text=True)
    assert plain.returncode == 0 and plain.stdout != r.stdout  # (another chain: the visit order is keyed on ids)


def test_cli_marginalize_matches_oracle_replay_and_the_python_driver():
    """`mcmc --marginalize` (the mode README.md:49-94 describes; -b / -f are dead flags in the reference): burn-in, samples
    every -f steps in whole sweeps, MAP label per node over samples and chains.  One compat chain replayed with the
    oracle; eight Philox chains against the Python driver on the same seeds."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    el = os.path.join(O.GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist")
    sizes = [125] * 4 + [84, 84, 83, 83, 83, 83]
    common = [cli, "-e", el, "-y", "500", "500", "-n", *map(str, sizes), "-z", "4", "6", "-b", "3000", "-t", "20000", "-f", "2000",
              "-E", "1", "-d", "3", "--gen_seed", "4", "--marginalize"]
    burn, between, samples, n = 3, 2, 10, 1000
    r = subprocess.run(common, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "10 samples 2 sweep(s) apart" in r.stderr
    rowptr, col = B.load_graph(el, n)
    o = O.OracleModel(rowptr, col, 500, 500, 4, 6, 1.0, O.labels_from_sizes(sizes))
    o.seed_compat(3, 4)
    o.init_bisbm()
    o.anneal("constant", [1.0], burn * n, BIG)
    want = np.zeros((n, 6), dtype=np.int64)
    for _ in range(samples):
        o.anneal("constant", [1.0], between * n, BIG)
        want += B.distributed.numpy_marginals(o.memberships()[None, :], 500, 4, 6)
    base = np.where(np.arange(n) >= 500, 4, 0)
    assert r.stdout == " ".join(map(str, want.argmax(axis=1) + base)) + " \n"
    # several Philox chains, randomised start: the Python driver on the same seeds pools the same histogram
    r8 = subprocess.run(common + ["--rng", "philox", "--chains", "8", "--randomize"], capture_output=True, text=True)
    assert r8.returncode == 0, r8.stderr
    g = gpu_model(rowptr, col, 500, 500, 4, 6, 1.0, O.labels_from_sizes(sizes), n_chains=8, rng="philox", seed=3)
    g.shuffle_bisbm()
    lab, counts = B.marginalize(g, burn, samples, between)
    assert counts.sum() == 8 * samples * n
    assert r8.stdout == " ".join(map(str, lab)) + " \n"
    # too few steps for one sample: an error, not an empty line
    bad = subprocess.run(common[:-1] + ["-t", "500", "--marginalize"], capture_output=True, text=True)
    assert bad.returncode == 1 and "no sample" in bad.stderr


def test_examples_run(golden):
    """examples/: the reference README's walk-through against the re-hosted CLI, and the Python mirror of the class API."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["bash", os.path.join(root, "examples", "maximization.sh")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    want = golden["compat_rng"]["scenario1_config1"]["labels"]
    assert r.stdout.split() == [str(x) for x in want]  # the reference's recorded line for these seeds
    r = subprocess.run(["bash", os.path.join(root, "examples", "marginalization.sh")], capture_output=True, text=True)
    assert r.returncode == 0 and len(r.stdout.split()) == 1000 and "256 chain(s) pooled" in r.stderr, r.stderr
    lab = np.array(r.stdout.split(), dtype=int)
    assert set(lab[:500]) <= set(range(4)) and set(lab[500:]) <= set(range(4, 10))
    r = subprocess.run(["bash", os.path.join(root, "examples", "estimate_k.sh")], capture_output=True, text=True)
    assert r.returncode == 0 and len(r.stdout.split()) == 1000 and "(Ka, Kb) = (4, 6) " in r.stderr, r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "python_api.py")], capture_output=True, text=True)
    assert r.returncode == 0 and "blocks now: 4 + 6" in r.stdout, r.stdout + r.stderr


# ------------------------------------------------------------------ full size: properties
def test_full_size_properties():
    """BASELINE configs[2] as benchmarked (N_a=N_b=5e5, E=1e7, Ka=Kb=32, 1024 chains): one sweep keeps the
    incremental state equal to a recount (sampled chains), block sizes sum to N, and sum dS equals the change of
    the full description length (every chain)."""
    na = nb = 500_000
    ka = kb = 32
    a, b = SYN.planted_edges(na, nb, 10_000_000, ka, kb, seed=1)
    rowptr, col = B.edge_to_adj((a, b), na + nb)
    labels = SYN.contiguous_labels(na, nb, ka, kb)
    chains = 1024
    g = gpu_model(rowptr, col, na, nb, ka, kb, 1.0, labels, n_chains=chains, rng="philox", seed=1)
    g.shuffle_bisbm()
    s0 = g.entropy()
    rates = B.MetropolisHasting().anneal(g, "constant", [1.0], na + nb, BIG)
    assert ((rates > 0.3) & (rates <= 1.0)).all()
    s1 = g.entropy()
    cum = g.get_entropy()
    assert np.allclose(s1 - s0, cum, rtol=1e-9, atol=1e-6 * np.abs(cum).max())
    picks = (0, 517, chains - 1)
    before = [(g.get_m(c), g.get_m_r(c), g.get_n_r(c), g.get_eta_rk_(c)) for c in picks]
    labs = [g.get_memberships(c) for c in picks]
    g.init_bisbm()  # recount from the labels
    for (m, m_r, n_r, eta), c, lab in zip(before, picks, labs):
        assert (g.get_m(c) == m).all() and (g.get_m_r(c) == m_r).all()
        assert (g.get_n_r(c) == n_r).all() and (g.get_eta_rk_(c) == eta).all()
        assert n_r.sum() == na + nb and m_r.sum() == 2 * 10_000_000
        assert (np.bincount(lab, minlength=ka + kb) == n_r).all()
    ms, updates = g.last_sweep_timing()
    assert updates == chains * (na + nb) and ms > 0

@pytest.mark.parametrize("mode", ["philox", "compat"])
def test_splits_cross_256_blocks_and_come_back(mode):
    """127 + 128 blocks (byte labels), three splits: 128 + 128 = 256 (still bytes), 128 + 129 = 257 (the labels become two bytes:
    wide mode), 129 + 129; a sweep in wide mode; a merge back to 120 + 120 (bytes again); a sweep.  Equal to the oracle after
    every call, in every chain."""
    rowptr, col = cases.random_graph(17, 400, 400, 6000, 127, 128)
    na = nb = 400
    n = na + nb
    labels = O.contiguous_labels(na, nb, 127, 128)
    chains = 3
    g = gpu_model(rowptr, col, na, nb, 127, 128, 1.0, labels, n_chains=chains, rng=mode, seed=21, gen_seed=22)
    g.shuffle_bisbm()
    os_ = []
    for c in range(chains):
        o = O.OracleModel(rowptr, col, na, nb, 127, 128, 1.0, labels)
        if mode == "compat":
            o.seed_compat(21 + c, 22 + c)
        else:
            o.seed_philox(21, c)
        o.shuffle_bisbm()
        os_.append(o)
    mh = B.MetropolisHasting()
    for da, db, shape in ((-1, 0, (128, 128)), (0, -1, (128, 129)), (-1, 0, (129, 129))):
        g.agg_merge(da, db, 6)
        assert (g.KA, g.KB) == shape
        for c, o in enumerate(os_):
            assert o.agg_merge(da, db, 6) == 0
            assert_state_equal(g, o, c)
    for sched, kw, (da, db) in (("constant", [1.0], (0, 0)), ("abrupt_cool", [0.0], (9, 9))):
        if da:
            g.agg_merge(da, db, 10)
            for o in os_:
                assert o.agg_merge(da, db, 10) == 0
        rg = mh.anneal(g, sched, kw, 2 * n, BIG)
        for c, o in enumerate(os_):
            assert o.anneal(sched, kw, 2 * n, BIG) == rg[c]
            assert_state_equal(g, o, c)
    assert (g.KA, g.KB) == (120, 120)
    ent = g.entropy()
    for c, o in enumerate(os_):
        assert abs(ent[c] - o.entropy()) <= 1e-9 * abs(o.entropy())

