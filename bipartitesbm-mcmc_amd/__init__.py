"""bipartitesbm-mcmc_amd -- host-side mirror of the reference's class API over the HIP C ABI.

The product is ``libbisbm_hip.so`` (csrc/, C ABI in include/bisbm.h): HIP kernels for gfx950 that
replace ``metropolis_hasting::anneal/step/transition_ratio`` and the hot part of ``blockmodel_t``
of junipertcy/bipartiteSBM-MCMC.  This module only binds it with ctypes and mirrors the names a
user of the reference knows (``blockmodel_t`` -> :class:`BlockModel`, ``metropolis_hasting`` ->
:class:`MetropolisHasting``, the five ``*_schedule`` functions, ``load_edge_list`` /
``edge_to_adj`` / ``load_memberships`` / ``output_vec``).

There is no CPU fallback: if the shared library cannot be loaded, importing the bound functions
raises, and ``bisbm_create`` fails with BISBM_ERR_NO_DEVICE when no HIP device is present.

The directory name contains a hyphen; load it with
``importlib.import_module("bipartitesbm-mcmc_amd")``.
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbisbm_hip.so")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

# --------------------------------------------------------------------------- enums of include/bisbm.h
BISBM_OK = 0
BISBM_ERR_INVALID_ARG = 1
BISBM_ERR_NOT_BIPARTITE = 2
BISBM_ERR_UNSUPPORTED = 3
BISBM_ERR_NO_DEVICE = 4
BISBM_ERR_HIP = 5
BISBM_ERR_STATE = 6
RNG_PHILOX = 0
RNG_MT19937_COMPAT = 1
ALL_CHAINS = -1
_RNG = {"philox": RNG_PHILOX, "mt19937-compat": RNG_MT19937_COMPAT, "compat": RNG_MT19937_COMPAT}

_u64p = C.POINTER(C.c_uint64)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_f64p = C.POINTER(C.c_double)
_f32p = C.POINTER(C.c_float)

# every symbol include/bisbm.h and include/bisbm_io.h declare: (restype, argtypes)
ABI = {
    "bisbm_abi_version": (C.c_int, []),
    "bisbm_check_shape": (C.c_int, [C.c_uint32, C.c_uint32, C.c_int]),
    "bisbm_create_multi": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint64, C.c_uint64, C.c_uint64, _u64p, _u32p, C.c_uint32, C.c_uint32,
                                     C.c_double, C.c_uint32, C.c_uint32, C.POINTER(C.c_int), C.c_int, C.c_int, C.c_uint64, C.c_uint64]),
    "bisbm_device_count": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), _u32p]),
    "bisbm_marginals_map": (C.c_int, [C.c_void_p, _u32p]),
    "bisbm_last_error": (C.c_char_p, [C.c_void_p]),
    "bisbm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint64, C.c_uint64, C.c_uint64, _u64p, _u32p,
                               C.c_uint32, C.c_uint32, C.c_double, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                               C.c_uint64, C.c_uint64]),
    "bisbm_destroy": (C.c_int, [C.c_void_p]),
    "bisbm_set_memberships": (C.c_int, [C.c_void_p, C.c_int64, _u32p]),
    "bisbm_init": (C.c_int, [C.c_void_p]),
    "bisbm_shuffle": (C.c_int, [C.c_void_p]),
    "bisbm_anneal": (C.c_int, [C.c_void_p, C.c_int, _f32p, C.c_uint64, C.c_uint64, _f64p]),
    "bisbm_get_memberships": (C.c_int, [C.c_void_p, C.c_uint32, _u32p]),
    "bisbm_get_block_state": (C.c_int, [C.c_void_p, C.c_uint32, _i32p, _i32p, _i32p, _u32p]),
    "bisbm_get_cum_dS": (C.c_int, [C.c_void_p, _f64p]),
    "bisbm_entropy": (C.c_int, [C.c_void_p, _f64p]),
    "bisbm_get_last_counts": (C.c_int, [C.c_void_p, _u64p, _u64p]),
    "bisbm_marginals_accumulate": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bisbm_marginals_reset": (C.c_int, [C.c_void_p]),
    "bisbm_marginals_get": (C.c_int, [C.c_void_p, _u32p]),
    "bisbm_get_ka_kb": (C.c_int, [C.c_void_p, _u32p, _u32p]),
    "bisbm_get_ka_kb_chain": (C.c_int, [C.c_void_p, C.c_uint32, _u32p, _u32p]),
    "bisbm_agg_merge": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "bisbm_agg_merge_total": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "bisbm_get_sizes": (C.c_int, [C.c_void_p, _u64p, _u64p, _u32p, _u32p]),
    "bisbm_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bisbm_last_sweep_timing": (C.c_int, [C.c_void_p, _f64p, _u64p]),
    "bisbm_last_pass_steps": (C.c_int, [C.c_void_p, _u32p]),
    "bisbm_debug_log_q": (C.c_int, [C.c_void_p, _i32p, _i32p, C.c_size_t, C.c_int, _f64p]),
    "bisbm_io_read_edge_list": (C.c_long, [C.c_char_p, C.POINTER(_u64p), C.POINTER(_u64p)]),
    "bisbm_io_read_memberships": (C.c_long, [C.c_char_p, C.POINTER(_u32p)]),
    "bisbm_io_edges_to_csr": (C.c_int, [_u64p, _u64p, C.c_size_t, C.c_uint64, _u64p, _u32p]),
    "bisbm_io_load_csr": (C.c_int, [C.c_char_p, C.c_uint64, C.c_int, C.POINTER(_u64p), C.POINTER(_u32p), _u64p,
                                    C.POINTER(C.c_int)]),
    "bisbm_io_locality_order": (C.c_int, [C.c_uint64, C.c_uint64, _u64p, _u32p, _u32p]),
    "bisbm_io_permute_csr": (C.c_int, [C.c_uint64, _u64p, _u32p, _u32p, _u64p, _u32p]),
    "bisbm_io_format_labels": (C.c_size_t, [_u32p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "bisbm_io_free": (None, [C.c_void_p]),
}

_lib = None


def build(force=False, verbose=False):
    """Compile csrc/ with hipcc for gfx950 into libbisbm_hip.so (in-tree)."""
    spec = importlib.util.spec_from_file_location("_bisbm_build", os.path.join(_HERE, "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.build(force=force, verbose=verbose)


def lib():
    """The loaded C-ABI library.  Raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libbisbm_hip.so is not built (run __graft_entry__.build() or "
                "`python bipartitesbm-mcmc_amd/build.py`); the engine has no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in ABI.items():
            fn = getattr(L, name)  # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class BisbmError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("bisbm error %d: %s" % (code, msg))
        self.code = code


def _p(a, t):
    return a.ctypes.data_as(t)


# --------------------------------------------------------------------------- cooling schedules
# metropolis_hasting.cc:10-37.  The reference passes a function pointer to anneal(); here the five
# functions are callables that also carry the id the kernel switches on.
class _Schedule:
    def __init__(self, name, sid):
        self.__name__ = name
        self.id = sid

    def __repr__(self):
        return "<%s>" % self.__name__


exponential_schedule = _Schedule("exponential_schedule", 0)
linear_schedule = _Schedule("linear_schedule", 1)
logarithmic_schedule = _Schedule("logarithmic_schedule", 2)
constant_schedule = _Schedule("constant_schedule", 3)
abrupt_cool_schedule = _Schedule("abrupt_cool_schedule", 4)
SCHEDULES = {
    "exponential": exponential_schedule, "linear": linear_schedule, "logarithmic": logarithmic_schedule,
    "constant": constant_schedule, "abrupt_cool": abrupt_cool_schedule,
}


def _schedule_id(s):
    if isinstance(s, _Schedule):
        return s.id
    if isinstance(s, str):
        return SCHEDULES[s].id
    return int(s)


# --------------------------------------------------------------------------- graph / membership I/O
def load_edge_list(path):
    """graph_utilities.cc:20-34 -> (a, b) uint64 arrays, one entry per line of the file."""
    L = lib()
    a, b = _u64p(), _u64p()
    n = L.bisbm_io_read_edge_list(os.fsencode(path), C.byref(a), C.byref(b))
    if n < 0:
        raise FileNotFoundError(path)
    ea = np.ctypeslib.as_array(a, shape=(max(n, 1),))[:n].copy()
    eb = np.ctypeslib.as_array(b, shape=(max(n, 1),))[:n].copy()
    L.bisbm_io_free(a)
    L.bisbm_io_free(b)
    return ea, eb


def load_memberships(path):
    """graph_utilities.cc:5-18 -> uint32 labels."""
    L = lib()
    p = _u32p()
    n = L.bisbm_io_read_memberships(os.fsencode(path), C.byref(p))
    if n < 0:
        raise FileNotFoundError(path)
    out = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].copy()
    L.bisbm_io_free(p)
    return out


def edge_to_adj(edge_list, num_vertices):
    """graph_utilities.cc:36-49, as CSR (rowptr uint64[n+1], col uint32[2E]); rows keep file order."""
    L = lib()
    a = np.ascontiguousarray(edge_list[0], dtype=np.uint64)
    b = np.ascontiguousarray(edge_list[1], dtype=np.uint64)
    rowptr = np.zeros(num_vertices + 1, dtype=np.uint64)
    col = np.zeros(max(2 * len(a), 1), dtype=np.uint32)
    rc = L.bisbm_io_edges_to_csr(_p(a, _u64p), _p(b, _u64p), len(a), num_vertices, _p(rowptr, _u64p),
                                 _p(col, _u32p))
    if rc != 0:
        raise ValueError("edge list has a node id >= %d" % num_vertices)
    return rowptr, col[: 2 * len(a)]


def load_graph(path, num_vertices, cache=False):
    """load_edge_list + edge_to_adj (graph_utilities.cc:20-49) in one call -> (rowptr, col).  cache=True keeps a binary
    CSR beside the text file (`<path>.bisbm_csr`, validated against the file's size and mtime and rebuilt when they
    change); the text format stays the source of truth.  `load_graph.last_cache_hit` tells where the arrays came from."""
    L = lib()
    rp, cl = _u64p(), _u32p()
    ne, hit = C.c_uint64(), C.c_int()
    rc = L.bisbm_io_load_csr(os.fsencode(path), int(num_vertices), int(bool(cache)), C.byref(rp), C.byref(cl), C.byref(ne),
                             C.byref(hit))
    if rc == -1:
        raise FileNotFoundError(path)
    if rc != 0:
        raise ValueError("edge list has a node id >= %d" % num_vertices)
    rowptr = np.ctypeslib.as_array(rp, shape=(num_vertices + 1,)).copy()
    col = np.ctypeslib.as_array(cl, shape=(max(2 * ne.value, 1),))[: 2 * ne.value].copy()
    L.bisbm_io_free(rp)
    L.bisbm_io_free(cl)
    load_graph.last_cache_hit = bool(hit.value)
    return rowptr, col


class LocalityOrder:
    """A renumbering of the nodes (type a within [0, na), type b within [na, n)) made by :func:`locality_order`:
    ``new_id[v]`` is the id node v has in the renumbered graph."""

    def __init__(self, new_id):
        self.new_id = np.ascontiguousarray(new_id, dtype=np.uint32)
        if len(self.new_id) and (self.new_id.max() >= len(self.new_id) or len(np.unique(self.new_id)) != len(self.new_id)):
            raise ValueError("new_id is not a permutation")
        self.old_id = np.empty_like(self.new_id)
        self.old_id[self.new_id] = np.arange(len(self.new_id), dtype=np.uint32)

    def apply(self, rowptr, col):
        """CSR of the renumbered graph (rows keep their edge order)."""
        L = lib()
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        rp = np.zeros_like(rowptr)
        cl = np.zeros(max(len(col), 1), dtype=np.uint32)
        if L.bisbm_io_permute_csr(len(rowptr) - 1, _p(rowptr, _u64p), _p(col, _u32p), _p(self.new_id, _u32p), _p(rp, _u64p),
                                  _p(cl, _u32p)) != 0:
            raise ValueError("bad permutation")
        return rp, cl[: len(col)]

    def to_new(self, per_node):
        """a per-node vector in the caller's numbering -> the engine's (e.g. initial memberships)"""
        return np.asarray(per_node)[self.old_id]

    def to_old(self, per_node):
        """a per-node vector from the engine -> the caller's numbering (e.g. get_memberships())"""
        return np.asarray(per_node)[self.new_id]


def locality_order(rowptr, col, na, nb):
    """Ingest-time renumbering for graphs whose ids carry no structure (include/bisbm_io.h): returns a LocalityOrder."""
    L = lib()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint64)
    col = np.ascontiguousarray(col, dtype=np.uint32)
    n = int(na) + int(nb)
    new_id = np.zeros(n, dtype=np.uint32)
    if L.bisbm_io_locality_order(n, int(na), _p(rowptr, _u64p), _p(col, _u32p), _p(new_id, _u32p)) != 0:
        raise ValueError("bad graph")
    return LocalityOrder(new_id)


def output_vec(vec, stream=None):
    """output_functions.hh:20-29: elements separated by blanks, trailing blank, newline."""
    L = lib()
    v = np.ascontiguousarray(vec, dtype=np.uint32)
    size = L.bisbm_io_format_labels(_p(v, _u32p), len(v), None, 0)
    buf = C.create_string_buffer(size + 2)
    L.bisbm_io_format_labels(_p(v, _u32p), len(v), buf, size + 2)
    text = buf.raw[:size].decode()
    (stream or sys.stderr).write(text)
    return text


# --------------------------------------------------------------------------- blockmodel_t
class BlockModel:
    """Mirror of ``blockmodel_t`` (blockmodel.hh:13-153) for ``n_chains`` independent chains.

    ``BlockModel(memberships, types, g, KA, KB, epsilon, adj)`` follows the reference constructor
    (blockmodel.hh:22-23): ``types`` is the 0/1 vector with all type-a nodes first (only the two
    counts are used), ``g`` is accepted and ignored like in the reference, ``adj`` is the CSR pair
    returned by :func:`edge_to_adj`.  Keyword extras select the chain-parallel parts the reference
    does not have.
    """

    def __init__(self, memberships, types, g, KA, KB, epsilon, adj, *, n_chains=1, rng="philox", seed=0,
                 gen_seed=0, device=0, first_chain_id=0, devices=None):
        """``devices``: a list of HIP device ordinals -- the chains are spread over them as contiguous ranges behind ONE handle
        (``bisbm_create_multi``); every result equals what a single device with all the chains gives."""
        L = lib()
        self._L = L
        types = np.asarray(types)
        self.na = int((types == 0).sum())
        self.nb = int((types == 1).sum())
        if self.na + self.nb != len(types) or (self.na and self.nb and types[: self.na].any()):
            raise ValueError("types must be 0 for the first NA nodes and 1 for the remaining NB")
        self.n = self.na + self.nb
        self.KA, self.KB = int(KA), int(KB)
        self.K = self.KA + self.KB
        self.mixed_shapes = False  # True once a one-argument agg_merge left the chains with different block counts
        self.epsilon = float(epsilon)
        self.n_chains = int(n_chains)
        self.device = int(device)
        rowptr = np.ascontiguousarray(adj[0], dtype=np.uint64)
        col = np.ascontiguousarray(adj[1], dtype=np.uint32)
        if len(rowptr) != self.n + 1:
            raise ValueError("adjacency has %d rows, types has %d nodes" % (len(rowptr) - 1, self.n))
        # A process that also uses torch.cuda (pooled marginals: a torch device tensor is handed to the library) must let
        # torch bring the device up first: its wheel carries its own HIP runtime, and the other order leaves torch
        # without a GPU.  Only done when the caller has imported torch already.
        _torch = sys.modules.get("torch")
        if _torch is not None:
            try:
                if _torch.cuda.is_available():
                    _torch.cuda.init()
            except Exception:
                pass
        h = C.c_void_p()
        self.devices = [int(d) for d in devices] if devices is not None else [int(device)]
        if devices is not None:
            self.device = self.devices[0]
            devs = (C.c_int * len(self.devices))(*self.devices)
            rc = L.bisbm_create_multi(C.byref(h), self.n, self.na, self.nb, _p(rowptr, _u64p), _p(col, _u32p), self.KA,
                                      self.KB, self.epsilon, self.n_chains, int(first_chain_id), devs, len(self.devices),
                                      _RNG[rng] if isinstance(rng, str) else int(rng), int(seed), int(gen_seed))
        else:
            rc = L.bisbm_create(C.byref(h), self.n, self.na, self.nb, _p(rowptr, _u64p), _p(col, _u32p), self.KA,
                                self.KB, self.epsilon, self.n_chains, int(first_chain_id), int(device),
                                _RNG[rng] if isinstance(rng, str) else int(rng), int(seed), int(gen_seed))
        if rc != BISBM_OK:
            raise BisbmError(rc, (L.bisbm_last_error(None) or b"").decode())
        self._h = h
        md, ne = C.c_uint32(), C.c_uint64()
        L.bisbm_get_sizes(h, None, C.byref(ne), C.byref(md), None)
        self.max_degree = md.value
        self.num_edges = ne.value
        self.set_memberships(memberships)

    # -- plumbing
    def _check(self, rc):
        if rc != BISBM_OK:
            raise BisbmError(rc, (self._L.bisbm_last_error(self._h) or b"").decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.bisbm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream):
        self._check(self._L.bisbm_set_stream(self._h, C.c_void_p(hip_stream)))

    # -- state
    def set_memberships(self, memberships, chain=ALL_CHAINS):
        mb = np.ascontiguousarray(memberships, dtype=np.uint32)
        if len(mb) != self.n:
            raise ValueError("memberships has %d entries, graph has %d nodes" % (len(mb), self.n))
        self._check(self._L.bisbm_set_memberships(self._h, int(chain), _p(mb, _u32p)))

    def init_bisbm(self):
        """blockmodel.cc:682-688"""
        self._check(self._L.bisbm_init(self._h))

    def shuffle_bisbm(self, engine=None, NA=None, NB=None):
        """blockmodel.cc:672-680 (the engine lives in the library; arguments kept for signature parity)"""
        self._check(self._L.bisbm_shuffle(self._h))

    # -- getters (blockmodel.cc:77-107)
    def get_memberships(self, chain=0):
        out = np.zeros(self.n, dtype=np.uint32)
        self._check(self._L.bisbm_get_memberships(self._h, int(chain), _p(out, _u32p)))
        return out

    def ka_kb(self, chain=0):
        """(KA, KB) of one chain: after a one-argument agg_merge the chains of a model may have different block counts."""
        ka, kb = C.c_uint32(), C.c_uint32()
        self._check(self._L.bisbm_get_ka_kb_chain(self._h, int(chain), C.byref(ka), C.byref(kb)))
        return ka.value, kb.value

    def _block_state(self, chain, want):
        K, D = sum(self.ka_kb(chain)), self.max_degree + 1
        m = np.zeros((K, K), dtype=np.int32) if "m" in want else None
        m_r = np.zeros(K, dtype=np.int32) if "m_r" in want else None
        n_r = np.zeros(K, dtype=np.int32) if "n_r" in want else None
        eta = np.zeros((K, D), dtype=np.uint32) if "eta" in want else None
        self._check(self._L.bisbm_get_block_state(
            self._h, int(chain), _p(m, _i32p) if m is not None else None,
            _p(m_r, _i32p) if m_r is not None else None, _p(n_r, _i32p) if n_r is not None else None,
            _p(eta, _u32p) if eta is not None else None))
        return m, m_r, n_r, eta

    def get_m(self, chain=0):
        return self._block_state(chain, ("m",))[0]

    def get_m_r(self, chain=0):
        return self._block_state(chain, ("m_r",))[1]

    def get_n_r(self, chain=0):
        return self._block_state(chain, ("n_r",))[2]

    def get_eta_rk_(self, chain=0):
        return self._block_state(chain, ("eta",))[3]

    def get_entropy(self):
        """Running sum of accepted dS per chain (blockmodel.cc:91)."""
        out = np.zeros(self.n_chains, dtype=np.float64)
        self._check(self._L.bisbm_get_cum_dS(self._h, _p(out, _f64p)))
        return out

    def entropy(self):
        """Full description length per chain (blockmodel.cc:753-787)."""
        out = np.zeros(self.n_chains, dtype=np.float64)
        self._check(self._L.bisbm_entropy(self._h, _p(out, _f64p)))
        return out

    def summary(self, stream=None):
        """blockmodel.cc:748-751"""
        s = stream or sys.stderr
        s.write("(Ka, Kb) = (%d, %d) \n" % (self.KA, self.KB))
        s.write("entropy: %s\n" % _fmt_g6(self.entropy()[0]))

    # -- agglomerative merges between anneals (blockmodel.cc:109-271)
    def _refresh_k(self):
        """KA / KB / K of the model: the counts all chains share, or -- once a one-argument agg_merge has left the chains
        with different ones (``mixed_shapes``) -- those of chain 0; ``ka_kb(chain)`` is always per chain."""
        ka, kb = C.c_uint32(), C.c_uint32()
        rc = self._L.bisbm_get_ka_kb(self._h, C.byref(ka), C.byref(kb))  # one call whatever the chain count:
        self.mixed_shapes = rc == BISBM_ERR_STATE                          # BISBM_ERR_STATE = the chains differ in shape
        if rc not in (BISBM_OK, BISBM_ERR_STATE):
            self._check(rc)
        self.KA, self.KB = self.ka_kb(0)
        self.K = self.KA + self.KB

    def agg_merge(self, diff_a, diff_b=None, nm=10):
        """``agg_merge(engine, diff_a, diff_b, nm)`` (blockmodel.cc:109-206) or, with ``diff_b=None``,
        ``agg_merge(engine, diff, nm)`` (:208-271), in every chain."""
        if diff_b is None:
            rc = self._L.bisbm_agg_merge_total(self._h, int(diff_a), int(nm))
        else:
            rc = self._L.bisbm_agg_merge(self._h, int(diff_a), int(diff_b), int(nm))
        self._check(rc)
        self._refresh_k()

    def get_KA(self):
        return self.KA

    def get_KB(self):
        return self.KB

    def get_num_edges(self):
        return self.num_edges

    def last_counts(self):
        acc = np.zeros(self.n_chains, dtype=np.uint64)
        sw = np.zeros(self.n_chains, dtype=np.uint64)
        self._check(self._L.bisbm_get_last_counts(self._h, _p(acc, _u64p), _p(sw, _u64p)))
        return acc, sw

    def last_sweep_timing(self):
        ms, upd = C.c_double(), C.c_uint64()
        self._check(self._L.bisbm_last_sweep_timing(self._h, C.byref(ms), C.byref(upd)))
        return ms.value, upd.value

    def last_pass_steps(self):
        """Steps per pass of the last sweep launch (1, 2, 4, 8): chosen per launch, never changes the chain."""
        k = C.c_uint32()
        self._check(self._L.bisbm_last_pass_steps(self._h, C.byref(k)))
        return k.value

    # -- marginals (README.md:49-53)
    @property
    def kmax(self):
        return max(self.KA, self.KB)

    def run_sweeps(self, sweeps, temperature=1.0):
        """`sweeps` sweeps at constant temperature (the "marginalize" regime: -c constant -a 1)."""
        return MetropolisHasting().anneal(self, constant_schedule, [temperature], int(sweeps) * self.n, 1 << 60)

    def counts_device(self):
        """torch device a caller-owned marginal histogram must live on."""
        import torch
        return torch.device("cuda", self.device)

    def marginals_reset(self):
        self._check(self._L.bisbm_marginals_reset(self._h))

    def marginals_accumulate(self, device_ptr=None):
        """One sample of every chain's labels into the internal histogram (device_ptr None) or ADDED to the caller's
        device buffer of n * kmax uint32 / int32 at `device_ptr` (e.g. torch_tensor.data_ptr())."""
        self._check(self._L.bisbm_marginals_accumulate(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def marginals_get(self):
        out = np.zeros((self.n, self.kmax), dtype=np.uint32)
        self._check(self._L.bisbm_marginals_get(self._h, _p(out, _u32p)))
        return out

    def marginals_map(self):
        """MAP block of every node from the internal histogram (most frequent block, ties -> the lowest), pooled over the
        handle's devices on the devices (``bisbm_marginals_map``: reduce-scatter -> argmax -> all-gather)."""
        out = np.zeros(self.n, dtype=np.uint32)
        self._check(self._L.bisbm_marginals_map(self._h, _p(out, _u32p)))
        return out

    def device_layout(self):
        """(device ordinals, first chain of each device) behind this handle."""
        nd = C.c_int()
        self._check(self._L.bisbm_device_count(self._h, C.byref(nd), None, None))
        devs, first = (C.c_int * nd.value)(), np.zeros(nd.value, dtype=np.uint32)
        self._check(self._L.bisbm_device_count(self._h, C.byref(nd), devs, _p(first, _u32p)))
        return list(devs), first.tolist()

    def debug_log_q(self, n, k, fast=False):
        n = np.ascontiguousarray(n, dtype=np.int32)
        k = np.ascontiguousarray(k, dtype=np.int32)
        out = np.zeros(len(n), dtype=np.float64)
        self._check(self._L.bisbm_debug_log_q(self._h, _p(n, _i32p), _p(k, _i32p), len(n), int(bool(fast)),
                                              _p(out, _f64p)))
        return out


def _fmt_g6(x):
    """What `std::clog << double` prints (6 significant digits, %g)."""
    return "%g" % x


# --------------------------------------------------------------------------- metropolis_hasting
class MetropolisHasting:
    """Mirror of ``metropolis_hasting`` (metropolis_hasting.hh:23-71)."""

    def anneal(self, blockmodel, cooling_schedule, cooling_schedule_kwargs, duration, steps_await, engine=None):
        """metropolis_hasting.cc:64-101.  Returns the acceptance rate: a float for one chain, an
        array for several."""
        kw = np.zeros(2, dtype=np.float32)
        vals = list(cooling_schedule_kwargs)[:2]
        kw[: len(vals)] = vals
        rates = np.zeros(blockmodel.n_chains, dtype=np.float64)
        blockmodel._check(blockmodel._L.bisbm_anneal(blockmodel._h, _schedule_id(cooling_schedule), _p(kw, _f32p),
                                                     int(duration), int(steps_await), _p(rates, _f64p)))
        return float(rates[0]) if blockmodel.n_chains == 1 else rates


metropolis_hasting = MetropolisHasting
blockmodel_t = BlockModel

from .distributed import ChainShard, shard_chains  # noqa: E402,F401
from .marginalize import marginalize  # noqa: E402,F401
