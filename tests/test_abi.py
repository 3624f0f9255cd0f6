"""CPU checks of the drop-in boundary: the C-ABI library loads without a GPU and exports every
symbol include/*.h declares; no compute is called here."""
import ctypes as C
import importlib
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = importlib.import_module("bipartitesbm-mcmc_amd")


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return set(re.findall(r"\b(bisbm_[a-z_0-9A-Z]+)\s*\(", text))


@pytest.fixture(scope="module")
def built():
    if not os.path.exists(B.LIB_PATH):
        B.build()
    return B.lib()


def test_every_declared_symbol_is_exported_and_bound(built):
    declared = _declared("bisbm.h") | _declared("bisbm_io.h")
    assert declared, "header parse failed"
    raw = C.CDLL(B.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(raw, name), "libbisbm_hip.so does not export %s" % name
    assert declared == set(B.ABI), (declared ^ set(B.ABI))
    assert built.bisbm_abi_version() == 3


def test_check_shape_states_the_wide_mode_limit(built):
    """bisbm_check_shape answers without a device: every shape up to 256 blocks, wide shapes while a chain's m_r / n_r / k_v
    histogram fit the LDS of a CU (10 bytes per block + 4 for the larger type + 16 KiB, + 9984 B of generator state in
    mt19937-compat mode), refused with the numbers beyond -- which is what bisbm_create and `mcmc --merge` report up front."""
    ok = lambda ka, kb, mode: built.bisbm_check_shape(ka, kb, mode)
    assert ok(1, 1, 0) == 0 and ok(128, 128, 1) == 0 and ok(500, 500, 0) == 0 and ok(7000, 7000, 0) == 0
    assert ok(7372, 7372, 0) == 0 and ok(7373, 7373, 0) == B.BISBM_ERR_UNSUPPORTED  # 10 K + 16384 <= 163840
    assert ok(6873, 6873, 1) == 0 and ok(6874, 6874, 1) == B.BISBM_ERR_UNSUPPORTED
    assert b"LDS" in built.bisbm_last_error(None)
    assert ok(40000, 30000, 0) == B.BISBM_ERR_UNSUPPORTED and b"two bytes" in built.bisbm_last_error(None)
    assert ok(0, 3, 0) == B.BISBM_ERR_INVALID_ARG and ok(3, 3, 7) == B.BISBM_ERR_INVALID_ARG
    # bisbm_create refuses the same shapes before it looks for a device
    big_n = 20000
    with pytest.raises(B.BisbmError) as e:
        B.BlockModel(np.arange(big_n, dtype=np.uint32), np.r_[np.zeros(10000), np.ones(10000)].astype(np.uint32), big_n, 10000, 10000, 1.0,
                     (np.zeros(big_n + 1, dtype=np.uint64), np.zeros(0, dtype=np.uint32)))
    assert e.value.code == B.BISBM_ERR_UNSUPPORTED and "LDS" in str(e.value)


def test_header_cites_the_reference_interface():
    text = open(os.path.join(ROOT, "include", "bisbm.h")).read()
    for cite in ("blockmodel.hh:22-23", "metropolis_hasting.hh:48-53", "blockmodel.cc:672-680",
                 "blockmodel.cc:682-688", "blockmodel.cc:753-787", "metropolis_hasting.cc:64-101"):
        assert cite in text


def test_no_device_fails_loudly(built):
    """Without a HIP device create must fail with NO_DEVICE (never fall back to a CPU path).  On a
    GPU box the same call succeeds, so the check is only made when no device is visible."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    rowptr = np.array([0, 1, 2], dtype=np.uint64)
    col = np.array([1, 0], dtype=np.uint32)
    with pytest.raises(B.BisbmError) as e:
        B.BlockModel([0, 1], [0, 1], 2, 1, 1, 1.0, (rowptr, col))
    assert e.value.code == B.BISBM_ERR_NO_DEVICE
    # argument validation happens before the device is touched
    with pytest.raises(B.BisbmError) as e:
        B.BlockModel([0, 0], [0, 0], 2, 1, 1, 1.0, (rowptr, col))
    assert e.value.code in (B.BISBM_ERR_NOT_BIPARTITE, B.BISBM_ERR_INVALID_ARG)


def test_product_does_not_touch_the_oracle():
    """The product package must not import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "bipartitesbm-mcmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower().replace("no cpu fallback", ""), f
                assert "orc_" not in text, f


def test_host_io_matches_oracle(built, tmp_path):
    import oracle_lib as O
    p = tmp_path / "q.el"
    p.write_text("0\t5\n1 6\n\n2   7\r\nabc def\n3\n4 8 junk\n  1\t9  \n")
    a, b = B.load_edge_list(str(p))
    oa, ob = O.load_edge_list(str(p))
    assert (a == oa).all() and (b == ob).all()
    for name, n in (("southernWomen.edgelist", 32), ("bisbm-n_1000-ka_4-kb_6.edgelist", 1000)):
        path = os.path.join(O.GOLDEN, name)
        a, b = B.load_edge_list(path)
        oa, ob = O.load_edge_list(path)
        assert (a == oa).all() and (b == ob).all()
        r1, c1 = B.edge_to_adj((a, b), n)
        r2, c2 = O.edge_to_csr(oa, ob, n)
        assert (r1 == r2).all() and (c1 == c2).all()
    mb = B.load_memberships(os.path.join(O.GOLDEN, "n_1000_membership.txt"))
    assert (mb == O.load_memberships(os.path.join(O.GOLDEN, "n_1000_membership.txt"))).all()
    assert B.output_vec([3, 0, 12, 7], stream=open(os.devnull, "w")) == O.format_vec([3, 0, 12, 7]) == "3 0 12 7 \n"
    with pytest.raises(FileNotFoundError):
        B.load_edge_list(str(tmp_path / "missing"))
    with pytest.raises(ValueError):
        B.edge_to_adj((np.array([0]), np.array([5])), 3)


def test_cli_flag_handling_matches_reference_messages(built):
    """The re-hosted `mcmc` shell: the reference's validation messages / exit codes (mcmc_main.cc:99-239) are
    produced before any device is touched."""
    import subprocess
    cli = os.path.join(ROOT, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    if not os.path.exists(cli):
        B.build(force=True)
    el = os.path.join(ROOT, "tests", "golden", "southernWomen.edgelist")

    def run(*args):
        r = subprocess.run([cli, *args], capture_output=True, text=True)
        return r.returncode, r.stdout, r.stderr
    rc, out, err = run()
    assert rc == 0 and "MCMC algorithms for the bipartiteSBM" in err and out == ""
    assert run("-y", "18", "14") == (1, "", "edge_list_path is required (-e flag)\n")
    assert run("-e", el) == (1, "", "types is required for bisbm mode (-y flag)\n")
    assert run("-e", el, "-y", "1", "2", "3") == (1, "", "Number of types must be equal to 2!\n")
    rc, out, err = run("-e", el, "-y", "18", "14", "-n", "4", "-z", "5", "5", "-c", "exponential", "-a", "10", "1.5")
    assert rc == 1 and "alpha must be in ]0,1[" in err and "alpha=1.5" in err
    rc, out, err = run("-e", el, "-y", "18", "14", "-c", "nonsense", "-a", "1")
    assert rc == 1 and err.startswith("Invalid cooling schedule.")
    rc, out, err = run("-e", el, "-y", "18", "14", "-z", "5", "5")
    assert rc == 1 and err == "n is required (-n flag) if one does not specify the membership of nodes\n"
    rc, out, err = run("-e", el, "-y", "18", "13", "-n", "18", "14", "-z", "1", "1")
    assert rc == 1 and "Types do not sum to the number of vertices!" in err
    rc, out, err = run("-e", el, "--bogus")
    assert rc == 1 and "unrecognised option" in err
    # --merge starts from one block per node: 1000 blocks run in the library's wide mode (two-byte labels), so on a box
    # without a GPU the run gets as far as bisbm_create; beyond what wide mode serves (bisbm_check_shape) it is refused with a
    # plain message
    big = os.path.join(ROOT, "tests", "golden", "bisbm-n_1000-ka_4-kb_6.edgelist")
    rc, out, err = run("-e", big, "-y", "500", "500", "-n", "500", "500", "-z", "4", "6", "--merge", "-c", "abrupt_cool", "-a", "50", "-t", "1000")
    assert (rc == 3 and "no HIP device" in err and out == "") or (rc == 0 and len(out.split()) == 1000)
    # the merge drivers go through the engine: without a device they fail loudly (there is no CPU path)
    rc, out, err = run("-e", el, "-y", "18", "14", "-n", "18", "14", "-z", "5", "5", "--merge", "-c", "abrupt_cool",
                       "-a", "50", "-t", "320")
    assert (rc == 3 and "no HIP device" in err and out == "") or (rc == 0 and len(out.split()) == 32)


def test_csr_cache_round_trip(built, tmp_path):
    """load_graph(cache=True): the binary CSR beside the text file equals what the text gives (quirky lines included),
    is reused only while the text file's size and mtime match, is rebuilt after an edit or a corrupted cache, and a
    run without the flag never creates it (graph_utilities.cc:20-49 stays the format)."""
    import shutil
    import time
    p = tmp_path / "g.el"
    p.write_text("0\t5\n1 6\n\n2   7\r\nabc def\n3\n4 8 junk\n  1\t9  \n")
    n = 10
    want = B.edge_to_adj(B.load_edge_list(str(p)), n)
    r0 = B.load_graph(str(p), n)                       # no cache asked for
    assert not os.path.exists(str(p) + ".bisbm_csr") and not B.load_graph.last_cache_hit
    r1 = B.load_graph(str(p), n, cache=True)           # builds it
    assert not B.load_graph.last_cache_hit and os.path.exists(str(p) + ".bisbm_csr")
    r2 = B.load_graph(str(p), n, cache=True)           # uses it
    assert B.load_graph.last_cache_hit
    for r in (r0, r1, r2):
        assert (r[0] == want[0]).all() and (r[1] == want[1]).all()
    B.load_graph(str(p), n + 1, cache=True)            # another vertex count: not this cache
    assert not B.load_graph.last_cache_hit
    B.load_graph(str(p), n, cache=True)
    assert not B.load_graph.last_cache_hit             # (the n + 1 run replaced the file)
    assert B.load_graph(str(p), n, cache=True) and B.load_graph.last_cache_hit
    # an edited text file invalidates the cache (size and mtime are part of the key)
    time.sleep(0.01)
    with open(p, "a") as f:
        f.write("2 9\n")
    r3 = B.load_graph(str(p), n, cache=True)
    assert not B.load_graph.last_cache_hit
    want3 = B.edge_to_adj(B.load_edge_list(str(p)), n)
    assert (r3[0] == want3[0]).all() and (r3[1] == want3[1]).all() and len(r3[1]) == len(want[1]) + 2
    # a truncated cache file is ignored and replaced
    c = str(p) + ".bisbm_csr"
    data = open(c, "rb").read()
    open(c, "wb").write(data[:-4])
    r4 = B.load_graph(str(p), n, cache=True)
    assert not B.load_graph.last_cache_hit and (r4[1] == want3[1]).all()
    assert open(c, "rb").read() == data
    # errors
    with pytest.raises(FileNotFoundError):
        B.load_graph(str(tmp_path / "missing"), 3, cache=True)
    with pytest.raises(ValueError):
        B.load_graph(str(p), 5, cache=True)
    # a directory that cannot be written to: no cache, same arrays
    ro = tmp_path / "ro"
    ro.mkdir()
    shutil.copy(p, ro / "g.el")
    os.chmod(ro, 0o555)
    try:
        r5 = B.load_graph(str(ro / "g.el"), n, cache=True)
        assert (r5[1] == want3[1]).all()
    finally:
        os.chmod(ro, 0o755)


def test_locality_order_is_a_type_preserving_bijection_that_localises(built):
    """bisbm_io_locality_order on a planted graph whose ids were scrambled: a deterministic bijection within each type,
    the renumbered CSR equals the CSR of the renumbered edge list (rows keep edge order), and the 64-byte label sectors
    a chunk of 64 consecutive ids gathers from drop back to about what the generator's own numbering needs."""
    syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
    na, nb, E, k = 40_000, 30_000, 700_000, 8
    a, b = syn.planted_edges(na, nb, E, k, k, seed=3)
    rs = np.random.default_rng(1)
    pa, pb = rs.permutation(na).astype(np.uint64), rs.permutation(nb).astype(np.uint64)
    a2, b2 = pa[a.astype(np.int64)], pb[(b - na).astype(np.int64)] + na
    a2[:5], b2[:5] = a2[5:10], b2[5:10]  # a few duplicate edges
    n = na + nb + 3                      # three isolated type-b nodes at the end
    rowptr, col = B.edge_to_adj((a2, b2), n)
    lo = B.locality_order(rowptr, col, na, nb + 3)
    assert sorted(lo.new_id[:na]) == list(range(na)) and sorted(lo.new_id[na:]) == list(range(na, n))
    assert (B.locality_order(rowptr, col, na, nb + 3).new_id == lo.new_id).all()
    assert (lo.to_old(lo.to_new(np.arange(n))) == np.arange(n)).all()
    rp2, cl2 = lo.apply(rowptr, col)
    a3, b3 = lo.new_id[a2.astype(np.int64)].astype(np.uint64), lo.new_id[b2.astype(np.int64)].astype(np.uint64)
    want = B.edge_to_adj((a3, b3), n)
    assert (rp2 == want[0]).all() and (cl2 == want[1]).all()

    def sectors_per_chunk(x, y):
        return len(np.unique((x // 64).astype(np.int64) * (1 << 32) + (y - na) // 64)) / (na / 64)
    natural, scrambled, ordered = sectors_per_chunk(a, b), sectors_per_chunk(a2, b2), sectors_per_chunk(a3, b3)
    assert scrambled > 1.5 * natural and ordered < 1.15 * natural, (natural, scrambled, ordered)
    with pytest.raises(ValueError):
        B.LocalityOrder(np.array([0, 1, 5], dtype=np.uint32))


def test_no_kernel_spills_vector_registers(built):
    """The build records the compiler's resource usage per kernel (build.py: -Rpass-analysis=kernel-resource-usage, no effect on
    the code).  No kernel of the library may spill a vector register (a spilled VGPR is a scratch round trip in a loop that is
    bound by dependent latencies; the cooling-schedule variants of round 2 had 111), and the production sweep kernel must
    exist in all its variants: {eta in LDS, eta window} x {constant, cooling schedule} x {two steps per pass with more than
    32 blocks, two with up to 32, four with up to 32, four with up to 16, eight with up to 8}."""
    import json
    build = importlib.import_module("bipartitesbm-mcmc_amd.build")
    usage = json.load(open(build.RESOURCES))
    sweep = {k: v for k, v in usage.items() if "sweep_fast_kernel" in k}
    assert len(sweep) == 20, sorted(sweep)
    assert len(usage) >= 40
    spilled = {k: v["VGPRs Spill"] for k, v in usage.items() if v.get("VGPRs Spill", 0) != 0}
    assert not spilled, spilled
    assert all(v["VGPRs"] <= 256 and v["Occupancy [waves/SIMD]"] >= 2 for v in sweep.values())


def test_bench_host_helpers():
    """bench.py's host-side helpers (no GPU): the limits cpu_baseline sizes its worker pool by, the build tag the profile
    summaries are matched with, and the algorithmic bytes per update of SURVEY 8(d)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    total, affinity, quota, mem = bench._cpu_limits()
    assert total >= 1 and 1 <= affinity <= total and (quota is None or quota > 0) and (mem is None or mem > 0)
    assert bench._worker_rss_bytes(1_000_000, 10_000_000) == pytest.approx(1.35e9, rel=0.05)  # (measured: 1.35 GB)
    assert bench.b_alg_per_update(1_000_000, 10_000_000) == 110.0 and bench.b_alg_per_update(4_000_000, 50_000_000) == 135.0
    tag = bench.build_tag()
    assert len(tag) == 12 and tag == bench.build_tag()
    # the three summaries the line quotes carry a build tag, and bench.py marks them when it is not the running build
    for name in ("r04_traffic.json", "r04_issue.json", "r04_steady_state.json"):
        j = bench.profile_json(name)
        assert j is not None and len(j.get("build", "")) == 12, name
