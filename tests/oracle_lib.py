"""ctypes binding of oracle/_build/liboracle.so (the CPU checker).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product
package."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN = os.path.join(ROOT, "tests", "golden")

SCHEDULES = {"exponential": 0, "linear": 1, "logarithmic": 2, "constant": 3, "abrupt_cool": 4}

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)


def build_oracle():
    """(Re)build the checker libraries when they are missing (gcc/g++ only; seconds).  BISBM_ORACLE_SO points the binding
    at another build of the same source (the sanitizer run of tests/test_sanitizers.py)."""
    if os.environ.get("BISBM_ORACLE_SO"):
        return os.environ["BISBM_ORACLE_SO"]
    so = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
    chk = os.path.join(ORACLE_DIR, "_build", "libstdcheck.so")
    src = os.path.join(ORACLE_DIR, "bisbm_oracle.c")
    if (not os.path.exists(so) or not os.path.exists(chk)
            or os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.run(["make", "-C", ORACLE_DIR, "_build/liboracle.so", "_build/libstdcheck.so"],
                       check=True, capture_output=True)
    return so


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build_oracle())
    L.orc_load_edge_list.restype = C.c_long
    L.orc_load_edge_list.argtypes = [C.c_char_p, C.POINTER(_u64p), C.POINTER(_u64p)]
    L.orc_load_memberships.restype = C.c_long
    L.orc_load_memberships.argtypes = [C.c_char_p, C.POINTER(_u32p)]
    L.orc_edge_to_csr.restype = C.c_int
    L.orc_edge_to_csr.argtypes = [_u64p, _u64p, C.c_size_t, C.c_size_t, _u64p, _u32p]
    L.orc_format_vec.restype = C.c_size_t
    L.orc_format_vec.argtypes = [_u32p, C.c_size_t, C.c_char_p, C.c_size_t]
    L.orc_init_tables.argtypes = [C.c_size_t, C.c_size_t]
    for name in ("orc_lgamma_fast", "orc_safelog_fast"):
        getattr(L, name).restype = C.c_double
        getattr(L, name).argtypes = [C.c_size_t]
    L.orc_log_q.restype = C.c_double
    L.orc_log_q.argtypes = [C.c_int, C.c_int]
    L.orc_log_q_approx.restype = C.c_double
    L.orc_log_q_approx.argtypes = [C.c_size_t, C.c_size_t]
    L.orc_log_q_philox.restype = C.c_double
    L.orc_log_q_philox.argtypes = [C.c_int, C.c_int]
    L.orc_q_cache_at.restype = C.c_double
    L.orc_q_cache_at.argtypes = [C.c_size_t, C.c_size_t]
    L.orc_spence.restype = C.c_double
    L.orc_spence.argtypes = [C.c_double]
    L.orc_lbinom_fast.restype = C.c_double
    L.orc_lbinom_fast.argtypes = [C.c_size_t, C.c_size_t]
    L.orc_lgamma_table.restype = _f64p
    L.orc_lgamma_table.argtypes = [C.POINTER(C.c_size_t)]
    L.orc_q_table.restype = _f64p
    L.orc_q_table.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    L.orc_schedule.restype = C.c_double
    L.orc_schedule.argtypes = [C.c_int, C.c_uint64, C.c_float, C.c_float]
    L.orc_philox4x32_10.argtypes = [_u32p, _u32p, _u32p]
    L.orc_philox_visit.restype = C.c_uint32
    L.orc_philox_visit.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_create.restype = C.c_void_p
    L.orc_create.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, _u64p, _u32p, C.c_size_t,
                             C.c_size_t, C.c_double, _u32p]
    L.orc_destroy.argtypes = [C.c_void_p]
    L.orc_seed_compat.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
    L.orc_seed_philox.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32]
    L.orc_set_memberships.argtypes = [C.c_void_p, _u32p]
    L.orc_init_bisbm.argtypes = [C.c_void_p]
    L.orc_shuffle_bisbm.argtypes = [C.c_void_p]
    L.orc_anneal.restype = C.c_double
    L.orc_anneal.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_uint64, C.c_uint64]
    L.orc_entropy.restype = C.c_double
    L.orc_entropy.argtypes = [C.c_void_p]
    L.orc_get_entropy.restype = C.c_double
    L.orc_get_entropy.argtypes = [C.c_void_p]
    L.orc_compute_dS_vertex.restype = C.c_double
    L.orc_compute_dS_vertex.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t]
    L.orc_transition_ratio.restype = C.c_double
    L.orc_transition_ratio.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, _f64p]
    L.orc_pair_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_double, _u64p]
    L.orc_depth_probe.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_int, _u64p]
    L.orc_propose_philox.restype = C.c_size_t
    L.orc_propose_philox.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_double]
    for name in ("orc_n", "orc_k", "orc_num_edges", "orc_max_degree"):
        getattr(L, name).restype = C.c_size_t
        getattr(L, name).argtypes = [C.c_void_p]
    for name in ("orc_last_accepted", "orc_last_sweeps", "orc_total_sweeps"):
        getattr(L, name).restype = C.c_uint64
        getattr(L, name).argtypes = [C.c_void_p]
    L.orc_get_memberships.argtypes = [C.c_void_p, _u32p]
    L.orc_get_m.argtypes = [C.c_void_p, _i32p]
    L.orc_get_m_r.argtypes = [C.c_void_p, _i32p]
    L.orc_get_n_r.argtypes = [C.c_void_p, _i32p]
    L.orc_get_eta.argtypes = [C.c_void_p, _u32p]
    L.orc_get_vlist.argtypes = [C.c_void_p, _u32p]
    L.orc_ka.restype = C.c_size_t
    L.orc_ka.argtypes = [C.c_void_p]
    L.orc_kb.restype = C.c_size_t
    L.orc_kb.argtypes = [C.c_void_p]
    L.orc_merge_dS.restype = C.c_double
    L.orc_merge_dS.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t]
    L.orc_agg_merge.restype = C.c_int
    L.orc_agg_merge.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.orc_last_split_dS.restype = C.c_double
    L.orc_last_split_dS.argtypes = [C.c_void_p]
    L.orc_agg_split.restype = C.c_int
    L.orc_agg_split.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_agg_merge_total.restype = C.c_int
    L.orc_agg_merge_total.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.orc_geospace.restype = C.c_size_t
    L.orc_geospace.argtypes = [C.c_long, C.c_long, C.c_long, C.c_long, C.c_double, C.POINTER(C.c_int),
                               C.POINTER(C.c_int), C.c_size_t]
    L.free = C.CDLL(None).free
    L.free.argtypes = [C.c_void_p]
    _lib = L
    return L


def _p(a, t):
    return a.ctypes.data_as(t)


def load_edge_list(path):
    L = lib()
    a, b = _u64p(), _u64p()
    n = L.orc_load_edge_list(path.encode(), C.byref(a), C.byref(b))
    if n < 0:
        raise FileNotFoundError(path)
    ea = np.ctypeslib.as_array(a, shape=(max(n, 1),))[:n].copy()
    eb = np.ctypeslib.as_array(b, shape=(max(n, 1),))[:n].copy()
    L.free(a)
    L.free(b)
    return ea, eb


def load_memberships(path):
    L = lib()
    p = _u32p()
    n = L.orc_load_memberships(path.encode(), C.byref(p))
    if n < 0:
        raise FileNotFoundError(path)
    out = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()
    L.free(p)
    return out


def edge_to_csr(a, b, n):
    L = lib()
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    rowptr = np.zeros(n + 1, dtype=np.uint64)
    col = np.zeros(max(2 * len(a), 1), dtype=np.uint32)
    rc = L.orc_edge_to_csr(_p(a, _u64p), _p(b, _u64p), len(a), n, _p(rowptr, _u64p), _p(col, _u32p))
    if rc != 0:
        raise ValueError("edge id >= n")
    return rowptr, col[: 2 * len(a)]


def format_vec(v):
    L = lib()
    v = np.ascontiguousarray(v, dtype=np.uint32)
    buf = C.create_string_buffer(12 * len(v) + 8)
    n = L.orc_format_vec(_p(v, _u32p), len(v), buf, len(buf))
    return buf.raw[:n].decode()


def contiguous_labels(na, nb, ka, kb):
    """floor(i*K/N) per type: the initial partition of the SURVEY section-4 scenarios."""
    la = (np.arange(na, dtype=np.int64) * ka) // na
    lb = ka + (np.arange(nb, dtype=np.int64) * kb) // nb
    return np.concatenate([la, lb]).astype(np.uint32)


def labels_from_sizes(sizes):
    """-n block sizes -> contiguous labels (mcmc_main.cc:302-326)."""
    return np.repeat(np.arange(len(sizes), dtype=np.uint32), sizes)


class OracleModel:
    """blockmodel_t + metropolis_hasting of the reference, restated (oracle/bisbm_oracle.c)."""

    def __init__(self, rowptr, col, na, nb, ka, kb, epsilon, labels):
        L = lib()
        self.L = L
        self.n = na + nb
        self.na, self.nb, self.ka, self.kb = na, nb, ka, kb
        self.K = ka + kb
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        labels = np.ascontiguousarray(labels, dtype=np.uint32)
        assert len(rowptr) == self.n + 1 and len(labels) == self.n
        self.h = L.orc_create(self.n, na, nb, _p(rowptr, _u64p), _p(col, _u32p), ka, kb,
                              float(epsilon), _p(labels, _u32p))
        if not self.h:
            raise ValueError("orc_create failed")
        self.maxdeg = L.orc_max_degree(self.h)
        self.num_edges = L.orc_num_edges(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def seed_compat(self, engine_seed, gen_seed):
        self.L.orc_seed_compat(self.h, engine_seed, gen_seed)

    def seed_philox(self, seed, chain_id):
        self.L.orc_seed_philox(self.h, seed, chain_id)

    def set_memberships(self, labels):
        labels = np.ascontiguousarray(labels, dtype=np.uint32)
        self.L.orc_set_memberships(self.h, _p(labels, _u32p))

    def init_bisbm(self):
        self.L.orc_init_bisbm(self.h)

    def shuffle_bisbm(self):
        self.L.orc_shuffle_bisbm(self.h)

    def anneal(self, schedule, kwargs, duration, steps_await):
        kw = list(kwargs) + [0.0, 0.0]
        return self.L.orc_anneal(self.h, SCHEDULES[schedule], kw[0], kw[1], duration, steps_await)

    def _refresh_k(self):
        self.ka, self.kb = self.L.orc_ka(self.h), self.L.orc_kb(self.h)
        self.K = self.ka + self.kb

    def merge_dS(self, r, s):
        """compute_dS(block_move_t), blockmodel.cc:335-372"""
        return self.L.orc_merge_dS(self.h, r, s)

    def agg_merge(self, diff_a, diff_b, nm):
        """blockmodel_t::agg_merge(engine, diff_a, diff_b, nm), blockmodel.cc:109-206"""
        rc = self.L.orc_agg_merge(self.h, diff_a, diff_b, nm)
        self._refresh_k()
        return rc

    def agg_split(self, type_b, nm):
        """blockmodel_t::agg_split(engine, type, nm), blockmodel.cc:505-565 (intended semantics)"""
        rc = self.L.orc_agg_split(self.h, int(bool(type_b)), nm)
        self._refresh_k()
        return rc

    def agg_merge_total(self, diff, nm):
        """blockmodel_t::agg_merge(engine, diff, nm), blockmodel.cc:208-271"""
        rc = self.L.orc_agg_merge_total(self.h, diff, nm)
        self._refresh_k()
        return rc

    def entropy(self):
        return self.L.orc_entropy(self.h)

    def get_entropy(self):
        return self.L.orc_get_entropy(self.h)

    def compute_dS_vertex(self, v, r, s):
        return self.L.orc_compute_dS_vertex(self.h, v, r, s)

    def transition_ratio(self, v, s):
        acc = C.c_double(0)
        dS = self.L.orc_transition_ratio(self.h, v, s, C.byref(acc))
        return dS, acc.value

    def pair_probe(self, sweeps, temperature=1.0):
        """Runs `sweeps` Philox sweeps (exactly as anneal does) and reports how often the second step of a pass of the
        production kernel stands: dict(steps, passes, second_stood, ...)."""
        out = (C.c_uint64 * 6)()
        self.L.orc_pair_probe(self.h, sweeps, temperature, out)
        return dict(steps=out[0], passes=out[1], second_stood=out[2], first_moved=out[3], row_clashes=out[4],
                    column_clashes=out[5])

    def depth_probe(self, sweeps, depth, temperature=1.0):
        """Like pair_probe for passes of up to `depth` steps: (steps, passes, [share of passes committing j + 1 steps])."""
        out = (C.c_uint64 * 12)()
        self.L.orc_depth_probe(self.h, sweeps, temperature, depth, out)
        return out[0], out[1], [out[2 + j] / max(out[1], 1) for j in range(depth)]

    def propose_philox(self, v, u_idx, u_R, u_tgt):
        return self.L.orc_propose_philox(self.h, v, u_idx, u_R, u_tgt)

    @property
    def last_accepted(self):
        return self.L.orc_last_accepted(self.h)

    @property
    def last_sweeps(self):
        return self.L.orc_last_sweeps(self.h)

    def memberships(self):
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.orc_get_memberships(self.h, _p(out, _u32p))
        return out

    def m(self):
        out = np.zeros((self.K, self.K), dtype=np.int32)
        self.L.orc_get_m(self.h, _p(out, _i32p))
        return out

    def m_r(self):
        out = np.zeros(self.K, dtype=np.int32)
        self.L.orc_get_m_r(self.h, _p(out, _i32p))
        return out

    def n_r(self):
        out = np.zeros(self.K, dtype=np.int32)
        self.L.orc_get_n_r(self.h, _p(out, _i32p))
        return out

    def eta(self):
        out = np.zeros((self.K, self.maxdeg + 1), dtype=np.uint32)
        self.L.orc_get_eta(self.h, _p(out, _u32p))
        return out

    def vlist(self):
        out = np.zeros(self.n, dtype=np.uint32)
        self.L.orc_get_vlist(self.h, _p(out, _u32p))
        return out


def load_graph(name):
    """The two shipped datasets (copied as data fixtures into tests/golden/)."""
    if name == "southernWomen":
        path, na, nb = os.path.join(GOLDEN, "southernWomen.edgelist"), 18, 14
    elif name == "n_1000":
        path, na, nb = os.path.join(GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist"), 500, 500
    else:
        raise KeyError(name)
    a, b = load_edge_list(path)
    rowptr, col = edge_to_csr(a, b, na + nb)
    return rowptr, col, na, nb


def geospace(start_a, end_a, start_b, end_b, ratio):
    """support/util.hh:99-145"""
    cap = 4096
    a, b = (C.c_int * cap)(), (C.c_int * cap)()
    n = lib().orc_geospace(start_a, end_a, start_b, end_b, float(ratio), a, b, cap)
    return list(a[:n]), list(b[:n])
