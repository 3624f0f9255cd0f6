"""CPU tests of the oracle (oracle/bisbm_oracle.c) against what pins it:
  * the reference outputs recorded in SURVEY.md (tests/golden/survey_known_answers.json),
  * the real libstdc++ 11 (oracle/stdcheck.cc),
  * the reference's Boost-free TUs compiled as they lie (oracle/_ref/libref.so, when present),
  * invariants of the chain (incremental state == recount, sum dS == entropy difference)."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import oracle_lib as O

u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f64p = C.POINTER(C.c_double)


def _model(graph, ka, kb, eps, labels=None):
    rowptr, col, na, nb = O.load_graph(graph)
    if labels is None:
        labels = O.contiguous_labels(na, nb, ka, kb)
    return O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)


# ------------------------------------------------------------------ RNG-free known answers
def test_known_answers_southern_women(golden):
    g = golden["rng_free"]["southernWomen"]
    m = _model("southernWomen", 5, 5, g["epsilon"], O.labels_from_sizes(g["block_sizes"]))
    m.init_bisbm()
    assert m.num_edges == g["num_edges"]
    assert m.entropy() == g["entropy"]
    assert list(m.m_r()) == g["m_r"]
    assert list(m.m()[0][5:8]) == g["m_row0_cols5_7"]
    assert list(m.m()[2][6:10]) == g["m_row2_cols6_9"]
    for v, r, s, want in g["compute_dS"]:
        assert m.compute_dS_vertex(v, r, s) == want
    mm = m.m()
    assert (mm == mm.T).all() and (mm[:5, :5] == 0).all() and (mm[5:, 5:] == 0).all()


def test_known_answers_n_1000(golden):
    g = golden["rng_free"]["n_1000"]
    labels = O.load_memberships(os.path.join(O.GOLDEN, g["membership"]))
    assert len(labels) == 1000
    m = _model("n_1000", 4, 6, g["epsilon"], labels)
    m.init_bisbm()
    assert m.entropy() == g["entropy"]
    assert list(m.m_r()) == g["m_r"]
    assert list(m.m()[0]) == g["m_row0"]
    for v, r, s, want in g["compute_dS"]:
        assert m.compute_dS_vertex(v, r, s) == want


def test_log_q_known_answers(golden):
    L = O.lib()
    L.orc_init_tables(1 << 20, 501)
    for n, k, want in golden["rng_free"]["q_cache_exp"]:
        L.orc_init_tables(0, k)
        assert math.exp(L.orc_q_cache_at(n, k)) == pytest.approx(want, rel=1e-12)
    for n, k, want in golden["rng_free"]["log_q"]:
        assert L.orc_log_q(n, k) == want
    # edge cases of log_q (support/int_part.hh:27-37)
    assert L.orc_log_q(0, 5) == 0 and L.orc_log_q(-3, 5) == 0 and L.orc_log_q(10, 0) == 0
    assert L.orc_log_q(7, 100) == L.orc_log_q(7, 7)
    assert L.orc_lgamma_fast(0) == math.inf and L.orc_lgamma_fast(1) == 0.0
    libm = C.CDLL("libm.so.6")
    libm.lgamma.restype = C.c_double
    libm.lgamma.argtypes = [C.c_double]
    assert L.orc_lgamma_fast(11) == libm.lgamma(11.0)  # glibc's, not CPython's own lgamma
    assert L.orc_safelog_fast(0) == 0.0 and L.orc_safelog_fast(7) == math.log(7)


def test_schedules():
    L = O.lib()
    L.orc_init_tables(1 << 12, 2)
    f32 = np.float32
    # exponential: float * pow(double(float), double(t)); underflows to exactly 0
    assert L.orc_schedule(0, 0, 10, 0.1) == 10.0
    assert L.orc_schedule(0, 2, 10, 0.1) == float(f32(10)) * float(f32(0.1)) ** 2
    assert L.orc_schedule(0, 323, 10, 0.1) > 0.0 and L.orc_schedule(0, 324, 10, 0.1) == 0.0
    # linear is evaluated in FP32
    assert L.orc_schedule(1, 3, 5.5, 0.1) == float(f32(5.5) - f32(0.1) * f32(3))
    # logarithmic: +inf while floor(t + k1) <= 1
    assert L.orc_schedule(2, 0, 2.0, 1.0) == math.inf
    assert L.orc_schedule(2, 2, 2.0, 1.0) == 2.0 / math.log(3)
    assert L.orc_schedule(3, 12345, 0.7, 0) == float(f32(0.7))
    assert L.orc_schedule(4, 9, 10, 0) == 1.0 and L.orc_schedule(4, 10, 10, 0) == 0.0


# ------------------------------------------------------------------ compat-mode chains vs SURVEY
def _run_compat(golden, key):
    g = golden["compat_rng"][key]
    m = _model(g["graph"], g["ka"], g["kb"], g["epsilon"])
    m.seed_compat(42, 43)
    m.shuffle_bisbm()
    return g, m


@pytest.mark.parametrize("key", ["sample_southernWomen", "sample_n_1000"])
def test_compat_first_sweep(golden, key):
    g, m = _run_compat(golden, key)
    assert m.entropy() == g["S0"]
    rate = m.anneal(g["schedule"], g["kwargs"], g["duration"], 1 << 60)
    assert rate == g["rate"]
    assert m.get_entropy() == g["sum_dS"]


def test_compat_scenario1_config1(golden):
    g, m = _run_compat(golden, "scenario1_config1")
    s_before = m.entropy()
    rate = m.anneal(g["schedule"], g["kwargs"], g["duration"], g["steps_await"])
    assert rate == g["rate"]
    assert m.last_sweeps == g["sweeps"] and m.last_accepted == g["accepted"]
    assert m.get_entropy() == g["sum_dS"]
    assert list(m.memberships()) == g["labels"]
    assert m.entropy() == pytest.approx(g["entropy_approx"], abs=1e-3)
    # transition_ratio's dS is exactly the change of the full description length
    assert m.entropy() - s_before == pytest.approx(m.get_entropy(), abs=1e-10)


def test_compat_scenario2(golden):
    g, m = _run_compat(golden, "scenario2_n1000_constant")
    rate = m.anneal(g["schedule"], g["kwargs"], g["duration"], 1 << 60)
    assert rate == g["rate"]
    assert m.get_entropy() == g["sum_dS"]


def test_compat_scenario3(golden):
    g, m = _run_compat(golden, "scenario3_n1000_abrupt")
    rate = m.anneal(g["schedule"], g["kwargs"], g["duration"], g["steps_await"])
    assert rate == g["rate"]
    assert m.get_entropy() == g["sum_dS"]


def test_anneal_splits_compose():
    """anneal(constant, N) S times == one anneal(constant, S*N)."""
    a = _model("n_1000", 4, 6, 1.0)
    b = _model("n_1000", 4, 6, 1.0)
    for m in (a, b):
        m.seed_compat(7, 8)
        m.shuffle_bisbm()
    a.anneal("constant", [1.0], 5000, 1 << 60)
    for _ in range(5):
        b.anneal("constant", [1.0], 1000, 1 << 60)
    assert (a.memberships() == b.memberships()).all()
    assert a.get_entropy() == b.get_entropy()


# ------------------------------------------------------------------ invariants, both RNG modes
def _recount(m):
    fresh = O.OracleModel(m._rowptr, m._col, m.na, m.nb, m.ka, m.kb, 1.0, m.memberships())
    fresh.init_bisbm()
    return fresh


@pytest.mark.parametrize("mode", ["compat", "philox"])
@pytest.mark.parametrize("graph,ka,kb,eps", [("southernWomen", 5, 5, 0.001), ("n_1000", 4, 6, 1.0)])
def test_incremental_state_matches_recount(mode, graph, ka, kb, eps):
    rowptr, col, na, nb = O.load_graph(graph)
    m = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, O.contiguous_labels(na, nb, ka, kb))
    m._rowptr, m._col = rowptr, col
    if mode == "compat":
        m.seed_compat(1, 2)
    else:
        m.seed_philox(1234, 5)
    m.shuffle_bisbm()
    s0 = m.entropy()
    m.anneal("constant", [1.0], 20 * (na + nb), 1 << 60)
    f = _recount(m)
    assert (m.m() == f.m()).all() and (m.m_r() == f.m_r()).all()
    assert (m.n_r() == f.n_r()).all() and (m.eta() == f.eta()).all()
    assert (m.n_r() > 0).all()
    assert m.entropy() - s0 == pytest.approx(m.get_entropy(), abs=1e-8)


def test_philox_reproducible_and_chain_dependent():
    def run(seed, chain):
        m = _model("n_1000", 4, 6, 1.0)
        m.seed_philox(seed, chain)
        m.shuffle_bisbm()
        r = m.anneal("constant", [1.0], 10000, 1 << 60)
        return r, m.memberships(), m.get_entropy()
    r0, l0, e0 = run(99, 0)
    r1, l1, e1 = run(99, 0)
    r2, l2, e2 = run(99, 1)
    assert r0 == r1 and (l0 == l1).all() and e0 == e1
    assert not (l0 == l2).all()
    # splitting a Philox anneal is also exact (counters continue across calls)
    m = _model("n_1000", 4, 6, 1.0)
    m.seed_philox(99, 0)
    m.shuffle_bisbm()
    for _ in range(10):
        m.anneal("constant", [1.0], 1000, 1 << 60)
    assert (m.memberships() == l0).all() and m.get_entropy() == e0


def test_philox_kat_and_visit_order():
    L = O.lib()

    def phx(c, k):
        c = np.array(c, dtype=np.uint32)
        k = np.array(k, dtype=np.uint32)
        o = np.zeros(4, dtype=np.uint32)
        L.orc_philox4x32_10(c.ctypes.data_as(u32p), k.ctypes.data_as(u32p), o.ctypes.data_as(u32p))
        return list(o)
    # Random123 known-answer vectors for philox4x32-10
    assert phx([0] * 4, [0] * 2) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert phx([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert phx([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]
    for na, nb in ((1, 1), (2, 1), (3, 5), (18, 14), (500, 500), (4097, 3)):
        p = [L.orc_philox_visit(7, 3, 11, na, nb, i) for i in range(na + nb)]
        assert sorted(p) == list(range(na + nb))
        assert all(v < na for v in p[:na]) and all(v >= na for v in p[na:])  # type a phase, then type b
    a = [L.orc_philox_visit(7, 3, 11, 500, 500, i) for i in range(1000)]
    b = [L.orc_philox_visit(7, 3, 12, 500, 500, i) for i in range(1000)]
    assert a != b
    # the order is local in node ids: 64 consecutive positions visit one cell of 64 consecutive ids, 4096 consecutive
    # positions one tile of 4096 ids (up to the cycle-walked tail of a class that is not a multiple of 4096)
    na = 3 * 4096 + 700
    p = np.array([L.orc_philox_visit(99, 0, 5, na, 64, i) for i in range(na)])
    assert sorted(p) == list(range(na))
    full = p[: 3 * 4096].reshape(-1, 64)
    inside = [(c // 64 == c[0] // 64).all() for c in full]          # cells stay together ...
    assert sum(inside) >= len(full) - 16                            # ... except where the padded tail walks in
    tiles = p[: 3 * 4096].reshape(-1, 4096) // 4096
    assert all(np.bincount(t).max() >= 4096 - 64 * 12 for t in tiles)
    assert len({int(np.bincount(t).argmax()) for t in tiles}) == 3  # and the tiles come in a permuted order


# ------------------------------------------------------------------ libstdc++ 11, draw for draw
@pytest.fixture(scope="module")
def stdlib():
    O.build_oracle()
    S = C.CDLL(os.path.join(O.ORACLE_DIR, "_build", "libstdcheck.so"))
    S.std_shuffle.restype = C.c_double
    return S


@pytest.mark.parametrize("seed", [0, 42, 2 ** 32 + 5, 123456789])
def test_mt19937_and_canonical_match_libstdcxx(stdlib, seed):
    L = O.lib()
    n = 5000
    want = np.zeros(n, dtype=np.uint32)
    stdlib.std_mt_raw(C.c_uint64(seed), C.c_size_t(n), want.ctypes.data_as(u32p))
    g = O.C.create_string_buffer(4 * 624 + 8)
    L.orc_mt_seed(g, C.c_uint64(seed))
    L.orc_mt_next.restype = C.c_uint32
    got = np.array([L.orc_mt_next(g) for _ in range(n)], dtype=np.uint32)
    assert (got == want).all()
    wantd = np.zeros(n, dtype=np.float64)
    stdlib.std_canonical(C.c_uint64(seed), C.c_size_t(n), wantd.ctypes.data_as(f64p))
    L.orc_mt_seed(g, C.c_uint64(seed))
    L.orc_mt_canonical.restype = C.c_double
    gotd = np.array([L.orc_mt_canonical(g) for _ in range(n)])
    assert (gotd == wantd).all()


@pytest.mark.parametrize("n", [1, 2, 3, 32, 1000, 65535, 65536, 70001])
def test_shuffle_matches_libstdcxx(stdlib, n):
    L = O.lib()
    L.orc_mt_canonical.restype = C.c_double
    for seed in (1, 42):
        want = np.zeros(n, dtype=np.uint32)
        wu = stdlib.std_shuffle(C.c_uint64(seed), C.c_size_t(n), C.c_size_t(3), want.ctypes.data_as(u32p))
        g = O.C.create_string_buffer(4 * 624 + 8)
        L.orc_mt_seed(g, C.c_uint64(seed))
        v = np.arange(n, dtype=np.uint32)
        for _ in range(3):
            L.orc_mt_shuffle_u32(g, v.ctypes.data_as(u32p), C.c_size_t(n))
        assert (v == want).all()
        assert L.orc_mt_canonical(g) == wu


@pytest.mark.parametrize("n", [2, 3, 7, 32, 1001, 65535, 65536, 70000])
def test_bool_shuffle_matches_libstdcxx(stdlib, n):
    """agg_split's std::shuffle of a vector<bool> (blockmodel.cc:541-543) restated as a byte shuffle."""
    L = O.lib()
    u8p = C.POINTER(C.c_uint8)
    for seed in (3, 99):
        want = np.zeros(n, dtype=np.uint8)
        stdlib.std_shuffle_bool(C.c_uint64(seed), C.c_size_t(n), C.c_size_t(4), want.ctypes.data_as(u8p))
        g = O.C.create_string_buffer(4 * 624 + 8)
        L.orc_mt_seed(g, C.c_uint64(seed))
        v = (np.arange(n) >= n // 2).astype(np.uint8)
        for _ in range(4):
            L.orc_mt_shuffle_u8(g, v.ctypes.data_as(u8p), C.c_size_t(n))
        assert (v == want).all()


def test_discrete_and_uniform_int_match_libstdcxx(stdlib):
    L = O.lib()
    L.orc_mt_discrete.restype = C.c_size_t
    L.orc_mt_lemire.restype = C.c_uint32
    rng = np.random.default_rng(0)
    for trial in range(20):
        k = int(rng.integers(2, 40))
        w = rng.integers(0, 5000, k).astype(np.int32)
        w[rng.integers(0, k)] = 0
        if w.sum() == 0:
            w[0] = 1
        draws = 500
        want = np.zeros(draws, dtype=np.uint64)
        stdlib.std_discrete(C.c_uint64(trial), w.ctypes.data_as(C.POINTER(C.c_int)), C.c_size_t(k),
                            C.c_size_t(draws), want.ctypes.data_as(u64p))
        g = O.C.create_string_buffer(4 * 624 + 8)
        L.orc_mt_seed(g, C.c_uint64(trial))
        got = [L.orc_mt_discrete(g, w.ctypes.data_as(C.POINTER(C.c_int)), C.c_size_t(k)) for _ in range(draws)]
        assert got == list(want)
    for rangev in (2, 3, 6, 1000, 2 ** 31 + 7, 2 ** 32 - 1):
        want = np.zeros(2000, dtype=np.uint32)
        stdlib.std_uniform_int(C.c_uint64(5), C.c_uint32(rangev), C.c_size_t(2000), want.ctypes.data_as(u32p))
        g = O.C.create_string_buffer(4 * 624 + 8)
        L.orc_mt_seed(g, C.c_uint64(5))
        got = np.array([L.orc_mt_lemire(g, C.c_uint32(rangev)) for _ in range(2000)], dtype=np.uint32)
        assert (got == want).all()


# ------------------------------------------------------------------ reference TUs compiled as they lie
REF_SO = os.path.join(O.ORACLE_DIR, "_ref", "libref.so")
needs_ref = pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref not built")


@needs_ref
def test_spence_matches_reference_tu():
    R = C.CDLL(REF_SO)
    R.ref_spence.restype = C.c_double
    R.ref_spence.argtypes = [C.c_double]
    L = O.lib()
    xs = np.concatenate([[0.0, 1.0, 0.5, 1.5, 2.0, 1e-300, 1e-16, 3.0, 1e6],
                         np.random.default_rng(1).uniform(0, 4, 2000),
                         np.exp(-np.random.default_rng(2).uniform(0, 60, 2000))])
    for x in xs:
        assert L.orc_spence(float(x)) == R.ref_spence(float(x))
    assert math.isnan(L.orc_spence(-1.0)) and math.isnan(R.ref_spence(-1.0))


@needs_ref
def test_edge_list_io_matches_reference_tu(tmp_path):
    R = C.CDLL(REF_SO)
    R.ref_edge_list_to_csr.restype = C.c_long
    R.ref_edge_list_raw.restype = C.c_long
    R.ref_load_memberships.restype = C.c_long
    R.ref_output_vec.restype = C.c_size_t
    quirky = tmp_path / "quirky.el"
    # tabs, blanks, CRLF, a blank line (duplicates the previous edge), a non-numeric line
    # (pushes (0, previous b)), a one-number line, trailing junk
    quirky.write_text("0\t5\n1 6\n\n2   7\r\nabc def\n3\n4 8 junk\n  1\t9  \n")
    files = [(str(quirky), 10),
             (os.path.join(O.GOLDEN, "southernWomen.edgelist"), 32),
             (os.path.join(O.GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist"), 1000)]
    for path, n in files:
        a, b = O.load_edge_list(path)
        ra = np.zeros(len(a) + 8, dtype=np.uint64)
        rb = np.zeros(len(a) + 8, dtype=np.uint64)
        ne = R.ref_edge_list_raw(path.encode(), ra.ctypes.data_as(u64p), rb.ctypes.data_as(u64p),
                                 C.c_size_t(len(ra)))
        assert ne == len(a)
        assert (ra[:ne] == a).all() and (rb[:ne] == b).all()
        rowptr, col = O.edge_to_csr(a, b, n)
        n_out, nnz = C.c_size_t(0), C.c_size_t(0)
        R.ref_edge_list_to_csr(path.encode(), C.c_size_t(n), None, None, C.byref(n_out), C.byref(nnz))
        assert n_out.value == n and nnz.value == 2 * len(a)
        rrow = np.zeros(n + 1, dtype=np.uint64)
        rcol = np.zeros(nnz.value, dtype=np.uint32)
        R.ref_edge_list_to_csr(path.encode(), C.c_size_t(n), rrow.ctypes.data_as(u64p),
                               rcol.ctypes.data_as(u32p), C.byref(n_out), C.byref(nnz))
        assert (rrow == rowptr).all() and (rcol == col).all()
    mpath = os.path.join(O.GOLDEN, "n_1000_membership.txt")
    mine = O.load_memberships(mpath)
    ref = np.zeros(2000, dtype=np.uint32)
    cnt = R.ref_load_memberships(mpath.encode(), ref.ctypes.data_as(u32p), C.c_size_t(2000))
    assert cnt == len(mine) and (ref[:cnt] == mine).all()
    v = np.array([3, 0, 12, 7], dtype=np.uint32)
    buf = C.create_string_buffer(64)
    k = R.ref_output_vec(v.ctypes.data_as(u32p), C.c_size_t(4), buf, C.c_size_t(64))
    assert buf.raw[:k].decode() == O.format_vec(v) == "3 0 12 7 \n"


def test_edge_list_quirks_without_reference(tmp_path):
    """Same quirks, expected values written out, so the check also runs where _ref is absent."""
    p = tmp_path / "q.el"
    p.write_text("0\t5\n1 6\n\n2   7\r\nabc def\n3\n4 8 junk\n")
    a, b = O.load_edge_list(str(p))
    assert list(a) == [0, 1, 1, 2, 0, 3, 4] and list(b) == [5, 6, 6, 7, 7, 7, 8]
    rowptr, col = O.edge_to_csr(np.array([0, 1, 0]), np.array([2, 2, 2]), 3)
    assert list(rowptr) == [0, 2, 3, 6] and list(col) == [2, 2, 2, 0, 1, 0]


# ------------------------------------------------------------------ agglomerative merges
def _edge_entropy(m_full, m_r):
    """The terms compute_dS(block_move_t) (blockmodel.cc:335-372) keeps: -sum lg(m_ab + 1) over the a x b quadrant
    plus sum lg(m_r + 1)."""
    libm = C.CDLL("libm.so.6")
    libm.lgamma.restype = C.c_double
    libm.lgamma.argtypes = [C.c_double]
    K = len(m_r)
    s = 0.0
    for i in range(K):
        for j in range(i + 1, K):
            s -= libm.lgamma(float(m_full[i, j] + 1))
        s += libm.lgamma(float(m_r[i] + 1))
    return s


def test_merge_dS_is_the_change_of_the_edge_terms():
    rowptr, col, na, nb = O.load_graph("n_1000")
    labels = O.contiguous_labels(na, nb, 5, 7)
    o = O.OracleModel(rowptr, col, na, nb, 5, 7, 1.0, labels)
    o.init_bisbm()
    before = _edge_entropy(o.m(), o.m_r())
    for r, s in [(3, 1), (11, 6), (4, 0)]:
        merged = labels.copy()
        merged[merged == r] = s
        merged[merged > r] -= 1  # keep the numbering compact
        ka2, kb2 = (4, 7) if r < 5 else (5, 6)
        o2 = O.OracleModel(rowptr, col, na, nb, ka2, kb2, 1.0, merged)
        o2.init_bisbm()
        want = _edge_entropy(o2.m(), o2.m_r()) - before
        assert o.merge_dS(r, s) == pytest.approx(want, rel=1e-10)
    assert o.merge_dS(2, 2) == math.inf and o.merge_dS(7, 2) == math.inf  # same block, cross type


@pytest.mark.parametrize("mode", ["compat", "philox"])
def test_agg_merge_properties(mode):
    """Block counts reached, labels compact with type-a blocks first and numbered by first appearance, the block
    state equal to a recount, deterministic for a fixed seed; the --nature overload and the refusals."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    labels = O.contiguous_labels(na, nb, 9, 11)

    def fresh():
        o = O.OracleModel(rowptr, col, na, nb, 9, 11, 1.0, labels)
        o.seed_compat(3, 4) if mode == "compat" else o.seed_philox(3, 4)
        o.shuffle_bisbm()
        return o
    o = fresh()
    assert o.agg_merge(4, 5, 10) == 0
    assert (o.ka, o.kb) == (5, 6)
    lab = o.memberships()
    assert set(lab[:na]) == set(range(5)) and set(lab[na:]) == set(range(5, 11))
    firsts = [int(np.argmax(lab == k)) for k in range(11)]
    assert firsts == sorted(firsts)  # renumbered in the order of first appearance (blockmodel.cc:585-590)
    o2 = O.OracleModel(rowptr, col, na, nb, 5, 6, 1.0, lab)
    o2.init_bisbm()
    assert (o.m() == o2.m()).all() and (o.m_r() == o2.m_r()).all() and (o.n_r() == o2.n_r()).all()
    p = fresh()
    p.agg_merge(4, 5, 10)
    assert (p.memberships() == lab).all()
    assert o.agg_merge_total(3, 10) == 0 and o.K == 8
    assert o.agg_merge(0, 0, 10) == 0 and o.K == 8
    assert o.agg_merge_total(-1, 10) == -2       # the one-argument overload has no split branch (blockmodel.cc:208-271)
    one = O.OracleModel(rowptr, col, na, nb, 1, 3, 1.0, O.contiguous_labels(na, nb, 1, 3))
    one.seed_compat(1, 2) if mode == "compat" else one.seed_philox(1, 2)
    one.init_bisbm()
    assert one.agg_merge(1, 0, 10) == -3         # the reference would recurse without end


def test_geospace_known_values():
    """support/util.hh:99-145, worked by hand: the larger drop steps by floor(start / ratio^i)."""
    assert O.geospace(18, 5, 14, 5, 1.5) == ([18, 12, 8, 5], [14, 11, 8, 5])
    assert O.geospace(10, 10, 10, 10, 1.01) == ([10], [10])
    a, b = O.geospace(500, 4, 500, 6, 1.01)
    assert a[0] == 500 and a[-1] == 4 and b[0] == 500 and b[-1] == 6 and len(a) == len(b)
    assert all(x >= y for x, y in zip(a, a[1:])) and all(x >= y for x, y in zip(b, b[1:]))
    assert O.geospace(5, 2, 9, 3, 1.0) == ([0], [0])  # ratio <= 1


def test_philox_mode_log_q_definition():
    """Philox mode iterates get_v to convergence for u = k/sqrt(n) >= 2.5 (what the HIP kernels evaluate in closed form):
    identical to the reference's evaluation elsewhere, and inside that range different only by what the reference's
    |dv| <= 1e-8 stop leaves (<= 4e-10 relative at u = 2.5, shrinking fast with u)."""
    L = O.lib()
    rng = np.random.default_rng(5)
    for n, k in [(5000, 40), (10000, 500), (20000, 5), (160000, 21), (10001, 150), (1000000, 2400)]:  # table, small k, u < 2.5
        assert L.orc_log_q_philox(n, k) == L.orc_log_q(n, k)
    worst = {}
    for _ in range(3000):
        n = int(rng.integers(10001, 30_000_000))
        u = float(rng.uniform(2.5, 40.0))
        k = int(round(u * np.sqrt(n)))
        a, b = L.orc_log_q_philox(n, k), L.orc_log_q(n, k)
        band = 2.5 if u < 4 else 4 if u < 6 else 6 if u < 8 else 8 if u < 10 else 10 if u < 13 else 13
        worst[band] = max(worst.get(band, 0.0), abs(a - b) / abs(b))
    assert worst[2.5] < 8e-10 and worst[4] < 2e-10 and worst[6] < 2e-11 and worst[8] < 2e-12 and worst[10] < 3e-14 and worst[13] < 2e-15
    assert worst[2.5] > 1e-11  # (the stop does leave something there: the two definitions are not the same function)


def test_first_order_closed_form_of_log_q_is_converged_from_u_18():
    """The production kernels evaluate log_q for u = k / sqrt(n) > 18 by the FIRST-order closed form of get_v's fixed point
    (bisbm_device.hpp, log_q_closed; the boundary sat at 24 until the end of round 4) with a 1e-7-accurate exponential, and
    13 <= u <= 18 to second order.  Restated here in double arithmetic (the formula, not the device code): from u = 17.5 on it
    stays within 1e-15 of the converged evaluation (the Philox-mode definition) even with x off by 1.5e-7 either way, and the
    second-order term it leaves out is below 1e-17 of the result from u = 18 on (4e-16 at u = 15) -- while below u = 13.5 the
    first-order form alone is NOT enough (the test has power)."""
    L = O.lib()
    C0, C1 = math.pi / math.sqrt(6), 3 / math.pi ** 2
    lfc = math.log(C0) - 1.5 * math.log(2) - math.log(math.pi)

    def closed(n, k, x):
        u, sq = k / math.sqrt(n), math.sqrt(n)
        eps = (C1 * C0 * u + C1) * x
        t2 = 2 * C0 * sq
        return ((lfc - math.log(n)) + t2) + (x * (u * u * 0.25 + k + 0.5) - (eps * t2 + eps))

    def second_order(n, k, x0):
        u, sq = k / math.sqrt(n), math.sqrt(n)
        a, e1, h, t2 = C0 * u, C1 * (C0 * u + 1), 1 + u * u / 2, 2 * C0 * sq
        return x0 * x0 * (h * (a * e1 / 2 + h / 4) - e1 * e1 * (1 + t2 / 2) - C1 * (a * a * e1 + a / 2 + 0.25 + t2 / 4))

    rng = np.random.default_rng(18)
    worst_hi = worst_d2 = worst_lo = 0.0
    for _ in range(2500):
        n = int(rng.integers(10001, 60_000_000))
        k = max(1, int(round(float(rng.uniform(17.5, 30.0)) * math.sqrt(n))))
        want, x = L.orc_log_q_philox(n, k), math.exp(-C0 * k / math.sqrt(n))
        for f in (1.0, 1 + 1.5e-7, 1 - 1.5e-7):
            worst_hi = max(worst_hi, abs(closed(n, k, x * f) - want) / abs(want))
    for _ in range(500):
        n = int(rng.integers(10001, 60_000_000))
        k = max(1, int(round(float(rng.uniform(18.0, 19.0)) * math.sqrt(n))))
        worst_d2 = max(worst_d2, abs(second_order(n, k, math.exp(-C0 * k / math.sqrt(n)))) / abs(L.orc_log_q_philox(n, k)))
    for _ in range(500):
        n = int(rng.integers(1_000_000, 60_000_000))
        k = max(1, int(round(float(rng.uniform(13.0, 13.5)) * math.sqrt(n))))
        x = math.exp(-C0 * k / math.sqrt(n))
        want = L.orc_log_q_philox(n, k)
        worst_lo = max(worst_lo, abs(closed(n, k, x) - want) / abs(want))
        assert abs(closed(n, k, x) + second_order(n, k, x) - want) <= 2e-15 * abs(want)  # (... and with the second-order term it is)
    assert worst_hi < 1e-15, worst_hi
    assert worst_d2 < 1e-17, worst_d2
    assert worst_lo > 2e-15, worst_lo


# ------------------------------------------------------------------ agg_split (blockmodel.cc:374-459,505-565)
def _split_entropy_terms(m_full, m_r, ka):
    """The part of entropy() that compute_dS(split) tracks: -sum lgamma(m_rs + 1) over r < s, + sum lgamma(m_r + 1)."""
    K = len(m_r)
    e = 0.0
    for r in range(K):
        for s in range(r + 1, K):
            e -= math.lgamma(m_full[r, s] + 1)
        e += math.lgamma(m_r[r] + 1)
    return e


@pytest.mark.parametrize("mode", ["compat", "philox"])
@pytest.mark.parametrize("type_b", [0, 1])
def test_agg_split_properties(mode, type_b):
    """One more block of the asked type; exactly one block loses floor/ceil half of its nodes to the new label (KA for
    type a -- every type-b label moves up by one --, K for type b); everything else keeps its label; the state equals
    a recount; and the chosen cut is the best of nm independent evaluations (dS re-derived from the edge terms)."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    ka, kb = 3, 4
    m = O.OracleModel(rowptr, col, na, nb, ka, kb, 1.0, O.contiguous_labels(na, nb, ka, kb))
    if mode == "compat":
        m.seed_compat(11, 12)
    else:
        m.seed_philox(11, 4)
    m.shuffle_bisbm()
    m.anneal("constant", [1.0], 2 * (na + nb), 1 << 60)
    before = m.memberships().copy()
    n_r0 = m.n_r().copy()
    e0 = _split_entropy_terms(m.m(), m.m_r(), ka)
    assert m.agg_split(type_b, 8) == 0
    assert (m.ka, m.kb) == (ka + (1 - type_b), kb + type_b)
    after = m.memberships()
    new_label = ka if not type_b else ka + kb
    moved = np.flatnonzero(after == new_label)
    src = set(before[moved])
    assert len(src) == 1
    r = src.pop()
    assert (r >= ka) == bool(type_b)
    assert len(moved) == n_r0[r] - n_r0[r] // 2
    others = np.setdiff1d(np.arange(na + nb), moved)
    shift = (before[others] >= ka).astype(np.uint32) if not type_b else 0
    assert (after[others] == before[others] + shift).all()
    m._rowptr, m._col = rowptr, col
    f = _recount(m)
    assert (m.m() == f.m()).all() and (m.m_r() == f.m_r()).all() and (m.n_r() == f.n_r()).all()
    assert (m.eta() == f.eta()).all()
    # the cut that won is no worse than a plain first-half / second-half cut of any block of the type would
    # typically be: at least its dS (the change of the edge terms of the description length) is finite and the
    # new block is connected to the graph
    e1 = _split_entropy_terms(m.m(), m.m_r(), m.ka)
    assert np.isfinite(e1 - e0) and m.m_r()[new_label] > 0
    assert O.lib().orc_last_split_dS(m.h) == pytest.approx(e1 - e0, rel=1e-9)  # compute_dS(split) == change of the edge terms
    # deterministic for a fixed seed
    m2 = O.OracleModel(rowptr, col, na, nb, ka, kb, 1.0, O.contiguous_labels(na, nb, ka, kb))
    m2.seed_compat(11, 12) if mode == "compat" else m2.seed_philox(11, 4)
    m2.shuffle_bisbm()
    m2.anneal("constant", [1.0], 2 * (na + nb), 1 << 60)
    assert m2.agg_split(type_b, 8) == 0 and (m2.memberships() == after).all()
    # splitting a partition with single-node blocks only is refused (the reference would add an empty block)
    tiny = O.OracleModel(*O.load_graph("southernWomen")[:2], 18, 14, 18, 14, 0.001, np.arange(32, dtype=np.uint32))
    tiny.seed_compat(1, 2)
    tiny.init_bisbm()
    assert tiny.agg_split(0, 5) == -3


def test_agg_merge_with_negative_diffs_splits_first():
    """agg_merge(engine, -1, +1, nm) = one agg_split of type a, then one merge among type b (blockmodel.cc:110-117)."""
    rowptr, col, na, nb = O.load_graph("n_1000")
    a = O.OracleModel(rowptr, col, na, nb, 3, 5, 1.0, O.contiguous_labels(na, nb, 3, 5))
    b = O.OracleModel(rowptr, col, na, nb, 3, 5, 1.0, O.contiguous_labels(na, nb, 3, 5))
    for m in (a, b):
        m.seed_philox(5, 0)
        m.shuffle_bisbm()
    assert a.agg_merge(-1, 1, 10) == 0
    assert b.agg_split(0, 10) == 0 and b.agg_merge(0, 1, 10) == 0
    assert (a.ka, a.kb) == (4, 4) == (b.ka, b.kb)
    assert (a.memberships() == b.memberships()).all()
