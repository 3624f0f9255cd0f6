"""Ties the production ("Philox mode") chain back to the reference's own arithmetic and distribution -- CPU only.

Philox mode keeps the reference's Markov chain but evaluates it differently: butterfly sums,
the 1/deg factor dropped from both Hastings sums, an integer inverse-CDF proposal, log_q on the converged get_v fixed
point.  The oracle has both evaluations (mt19937-compat = the reference's, line for line; Philox = the production
definition the HIP kernels match bit for bit), so the bridge can be checked here without a GPU:

  * per step: dS and accu1/accu0 of transition_ratio (metropolis_hasting.cc:103-192) in both arithmetics, for every
    node and every admissible target of seeded states, with the tolerance stated;
  * per proposal: the exact target distribution of the integer inverse CDF against
    R_t/K + (1 - R_t) m[t][s]/m_r[t] (blockmodel.cc:619-628);
  * per chain: on a graph small enough to enumerate, many independent chains of either mode against exp(-S),
    S = entropy() (blockmodel.cc:753-787): chi-square on the state histogram.
"""
import numpy as np
import pytest

import cases
import oracle_lib as O


def _pair(rowptr, col, na, nb, ka, kb, eps, sweeps, seed=5):
    """The same seeded state in a compat-mode and a Philox-mode oracle model (after `sweeps` compat sweeps at T = 1)."""
    labels = O.contiguous_labels(na, nb, ka, kb)
    ref = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
    ref.seed_compat(seed, seed + 1)
    ref.shuffle_bisbm()
    if sweeps:
        ref.anneal("constant", [1.0], sweeps * (na + nb), 1 << 60)
    phx = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
    phx.seed_philox(99, 0)
    phx.set_memberships(ref.memberships())
    phx.init_bisbm()
    assert (phx.m() == ref.m()).all()
    return ref, phx


CROSS = [
    # name, sweeps before the comparison
    ("dense_low_tier", 0), ("dense_low_tier", 2),   # log_q tier 2.5 <= u < 8: where Philox mode departs most
    ("mid_tier_low", 0), ("mid_tier_low", 2),       # 8 <= u <= 24
    ("mid_tier", 1),                                # u around 23
    ("direct_tier", 1),                             # u > 24: closed form
    ("big_m_r", 1),                                 # literal tier (u < 2.5)
    ("hubs_isolated", 2),                           # table tier, degrees 0..150
]
# Tolerances of Philox-mode arithmetic against the reference's, per step:
#   |dS_phx - dS_ref| <= 1e-9 |S|   (S = description length; BASELINE north_star's 1e-9 relative is on the DL)
#   |dS_phx - dS_ref| <= 5e-6       absolute: what the converged get_v fixed point (vs the reference's |dv| <= 1e-8 stop)
#                                    can leave in four log_q terms of size 1e2..1e4; summation order alone gives ~1e-10
#   accu1/accu0 equal to 1e-12 relative
TOL_DS_REL_S = 1e-9
TOL_DS_ABS = 5e-6
TOL_ACCU = 1e-12


@pytest.mark.parametrize("name,sweeps", CROSS, ids=["%s-%d" % c for c in CROSS])
def test_philox_arithmetic_against_reference_arithmetic(name, sweeps, record_property):
    _, na, nb, ne, ka, kb, eps, hubs, iso = cases.CASE[name]
    rowptr, col = cases.random_graph(11, na, nb, ne, ka, kb, hubs, iso)
    ref, phx = _pair(rowptr, col, na, nb, ka, kb, eps, sweeps)
    S = abs(ref.entropy())
    lab = ref.memberships()
    n_r = ref.n_r()
    rng = np.random.default_rng(1)
    nodes = np.arange(na + nb) if na + nb <= 12000 else np.sort(rng.choice(na + nb, 12000, replace=False))
    worst_abs, worst_rel, worst_acc, pairs = 0.0, 0.0, 0.0, 0
    for v in nodes:
        r = int(lab[v])
        lo, hi = (0, ka) if v < na else (ka, ka + kb)
        for s in range(lo, hi):
            if s == r:
                continue
            d_ref, a_ref = ref.transition_ratio(int(v), s)
            d_phx, a_phx = phx.transition_ratio(int(v), s)
            assert np.isfinite(d_ref) and np.isfinite(d_phx)
            err = abs(d_phx - d_ref)
            worst_abs = max(worst_abs, err)
            if abs(d_ref) > 1e-3:
                worst_rel = max(worst_rel, err / abs(d_ref))
            if rowptr[v + 1] > rowptr[v]:
                worst_acc = max(worst_acc, abs(a_phx - a_ref) / abs(a_ref))
            else:
                assert a_ref == 1.0 and a_phx == 1.0  # deg == 0 (:185-189)
            pairs += 1
    record_property("pairs", pairs)
    record_property("worst_abs_dS", worst_abs)
    record_property("worst_rel_dS", worst_rel)
    record_property("worst_rel_accu", worst_acc)
    print("%s after %d sweeps: %d (node, target) pairs, |S| = %.6g, max |dS_phx - dS_ref| = %.3g (%.3g |S|), "
          "max relative to |dS| = %.3g, max relative accu1/accu0 difference = %.3g"
          % (name, sweeps, pairs, S, worst_abs, worst_abs / S, worst_rel, worst_acc))
    assert pairs > 0
    assert worst_abs <= TOL_DS_REL_S * S
    assert worst_abs <= TOL_DS_ABS
    assert worst_acc <= TOL_ACCU
    assert n_r.min() >= 1


@pytest.mark.parametrize("graph,ka,kb,eps", [("southernWomen", 5, 5, 0.001), ("n_1000", 4, 6, 1.0)])
def test_philox_proposal_distribution_is_the_references(graph, ka, kb, eps):
    """The integer inverse-CDF proposal enumerated over its uniforms: pivot = floor(u1 deg) hits every adjacency entry
    once; the random-target test flips at u2 = R_t = eps K / (m_r[t] + eps K) (:622) and then floor(u3 K) hits every one
    of the K blocks once (wrong-type blocks included, as in the reference, :624); otherwise x = floor(u3 m_r[t])
    over x = 0..m_r[t]-1 lands on own-type block s exactly m[t][s] times (:627-628: discrete_distribution(m_[t])).
    With 53-bit uniforms each of those cells has probability equal to its width up to 2^-53, so
    P(s | v) = sum_j 1/deg (R_t/K + (1 - R_t) m[t][s]/m_r[t]), t = b[j], j over the adjacency entries of v."""
    rowptr, col, na, nb = O.load_graph(graph)
    ref, phx = _pair(rowptr, col, na, nb, ka, kb, eps, 1)
    K = ka + kb
    lab, m, m_r = phx.memberships(), phx.m(), phx.m_r()
    rng = np.random.default_rng(3)
    checked_cdf = 0
    for v in rng.choice(na + nb, 12, replace=False):
        v = int(v)
        deg = int(rowptr[v + 1] - rowptr[v])
        if deg == 0:
            continue
        own = range(0, ka) if v < na else range(ka, K)
        for which in sorted(set(int(x) for x in rng.choice(deg, min(deg, 3), replace=False))):
            u1 = (which + 0.5) / deg
            t = int(lab[col[rowptr[v] + which]])
            R_t = eps * K / (m_r[t] + eps * K)
            # uniform branch: every block of either type once
            hit = [phx.propose_philox(v, u1, R_t * 0.5, (s + 0.5) / K) for s in range(K)]
            assert hit == list(range(K))
            # the test flips at R_t
            below = phx.propose_philox(v, u1, R_t * (1 - 1e-12), 1 - 1e-9)
            assert below == K - 1  # floor(u3 K) with u3 just under 1
            # inverse CDF branch: counts per target == row m[t][.]
            tot = int(m_r[t])
            counts = np.zeros(K, dtype=np.int64)
            for x in range(tot):
                counts[phx.propose_philox(v, u1, min(R_t * (1 + 1e-12), 1.0), (x + 0.5) / tot)] += 1
            assert (counts == m[t]).all(), (v, which, t)
            assert counts[[s for s in range(K) if s not in own]].sum() == 0
            checked_cdf += tot
    assert checked_cdf > 0


# ------------------------------------------------------------------ stationary distribution on an enumerable graph
N_CHAINS = 24000
BURN_IN = 40
N_CHAINS_PHILOX = 24000
BURN_IN_PHILOX = 40


def _oracle_samples(mode, n_chains, burn_in, T=1.0):
    rowptr, col = cases.enumerable_graph()
    na, nb = cases.ENUM_NA, cases.ENUM_NB
    start = O.contiguous_labels(na, nb, 2, 2)
    o = O.OracleModel(rowptr, col, na, nb, 2, 2, cases.ENUM_EPS, start)
    codes = np.zeros(n_chains, dtype=np.int64)
    for c in range(n_chains):
        o.set_memberships(start)
        if mode == "compat":
            o.seed_compat(1000 + c, 500000 + c)
        else:
            o.seed_philox(4242, c)
        o.shuffle_bisbm()
        o.anneal("constant", [T], burn_in * (na + nb), 1 << 60)
        codes[c] = cases.state_code(o.memberships())
    return codes


@pytest.fixture(scope="module")
def enum_pi():
    return cases.enumerable_states()


def test_enumerable_graph_is_a_real_test_case(enum_pi):
    states, prob, S = enum_pi
    assert len(states) == 3844 and abs(prob.sum() - 1) < 1e-12
    # not a near-uniform or near-degenerate distribution: a wrong chain shows up
    assert 0.002 < prob.max() < 0.2
    assert (prob * N_CHAINS >= 8).sum() > 150


@pytest.mark.parametrize("mode", ["compat", "philox"])
def test_chains_sample_exp_minus_S(mode, enum_pi):
    """Independent chains (one sample each, after a burn-in from a randomised start) against exp(-S): the reference's
    arithmetic (compat) and the production definition (philox) have the same stationary distribution."""
    states, prob, S = enum_pi
    codes = _oracle_samples(mode, N_CHAINS if mode == "compat" else N_CHAINS_PHILOX,
                            BURN_IN if mode == "compat" else BURN_IN_PHILOX)
    stat, dof, p = cases.chi_square(codes, states, prob)
    print("%s: chi2 = %.1f on %d dof, p = %.3g" % (mode, stat, dof, p))
    assert p > 1e-3, (stat, dof, p)
    # power: the same samples reject a slightly wrong target (T = 1.15 instead of 1)
    w = np.exp(-(S - S.min()) / 1.15)
    stat2, dof2, p2 = cases.chi_square(codes, states, w / w.sum())
    assert p2 < 1e-6, (stat2, dof2, p2)


def test_chains_at_another_temperature_sample_exp_minus_S_over_T(enum_pi):
    """The same at constant T = 2 in the reference's arithmetic: a = -dS / T + log(accu_r) (metropolis_hasting.cc:52-57)
    targets exp(-S / T); tests/test_gpu_scale.py runs T = 2 and T = 0.6 on the production kernel."""
    states, _, S = enum_pi
    codes = _oracle_samples("compat", 24000, 40, T=2.0)
    target = np.exp(-(S - S.min()) / 2.0)
    stat, dof, p = cases.chi_square(codes, states, target / target.sum())
    print("compat, T = 2: chi2 = %.1f on %d dof, p = %.3g" % (stat, dof, p))
    assert p > 1e-3, (stat, dof, p)
    w = np.exp(-(S - S.min()) / 2.5)  # power: T = 2.5 is rejected by the same samples
    assert cases.chi_square(codes, states, w / w.sum())[2] < 1e-6


# ------------------------------------------------------------------ the rule behind "two steps per pass" 
def test_second_step_of_a_pass_stands_when_the_rule_says_so():
    """The production kernel evaluates steps q and q+1 against the state before step q and keeps step q+1's evaluation
    unless step q moved its node and (a) the two steps share a block, or (b) step q has edges to the block t' of step
    q+1's pivot AND step q+1's target lies strictly between step q's two blocks.  orc_pair_probe walks the chain
    serially, applies that rule, and aborts if a proposal it declared untouched differs after step q -- so reaching
    the end proves the rule on this run; the chain it produced must be the chain anneal() produces."""
    na = nb = 6000
    rowptr, col = cases.random_graph(4, na, nb, 120_000, 16, 16)
    lab = O.contiguous_labels(na, nb, 16, 16)
    probe = O.OracleModel(rowptr, col, na, nb, 16, 16, 1.0, lab)
    plain = O.OracleModel(rowptr, col, na, nb, 16, 16, 1.0, lab)
    for m in (probe, plain):
        m.seed_philox(3, 1)
        m.shuffle_bisbm()
    st = probe.pair_probe(3)
    plain.anneal("constant", [1.0], 3 * (na + nb), 1 << 60)
    assert (probe.memberships() == plain.memberships()).all() and (probe.m() == plain.m()).all()
    assert st["steps"] == 3 * (na + nb)
    stood = st["second_stood"] / st["passes"]
    print("passes %d, second step stood in %.1f %%, %.2f steps per pass" % (st["passes"], 100 * stood, st["steps"] / (st["passes"] + (st["steps"] - st["passes"] - st["second_stood"]))))
    assert 0.5 < stood < 1.0 and st["row_clashes"] > 0 and st["column_clashes"] > 0
    # the same rule composed over deeper passes (what a four-steps-per-pass kernel would rest on): still the serial chain
    deep = O.OracleModel(rowptr, col, na, nb, 16, 16, 1.0, lab)
    deep.seed_philox(3, 1)
    deep.shuffle_bisbm()
    steps, passes, share = deep.depth_probe(3, 4)
    assert steps == 3 * (na + nb) and (deep.memberships() == plain.memberships()).all()
    assert abs(sum(share) - 1) < 1e-9 and steps / passes > st["steps"] / (st["passes"] + (st["steps"] - st["passes"] - st["second_stood"]))
