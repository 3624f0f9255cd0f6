"""Several devices behind ONE handle of the C ABI (bisbm_create_multi, `mcmc --devices`), rehearsed on a one-GPU box by
listing device 0 more than once: contiguous chain ranges per device entry, global chain ids keying the streams, one host
thread and one stream per entry, the marginal histogram pooled on the device(s) by node range (RCCL refuses two ranks on one
device, so the rehearsal takes the peer-copy path of the same exchange).  Everything a multi-device handle returns must equal
what a single-device handle with all the chains returns -- and hence the oracle."""
import importlib
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O

pytestmark = pytest.mark.gpu

B = importlib.import_module("bipartitesbm-mcmc_amd")
SYN = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
BIG = 1 << 60
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(rowptr, col, na, nb, ka, kb, eps, labels, **kw):
    return B.BlockModel(labels, SYN.types_vector(na, nb), ka + kb, ka, kb, eps, (rowptr, col), **kw)


def _same_state(a, b, chains):
    for c in range(chains):
        assert a.ka_kb(c) == b.ka_kb(c), c
        assert (a.get_memberships(c) == b.get_memberships(c)).all(), c
        assert (a.get_m(c) == b.get_m(c)).all() and (a.get_m_r(c) == b.get_m_r(c)).all(), c
        assert (a.get_n_r(c) == b.get_n_r(c)).all() and (a.get_eta_rk_(c) == b.get_eta_rk_(c)).all(), c
    assert (np.atleast_1d(a.get_entropy()) == np.atleast_1d(b.get_entropy())).all()
    assert (np.atleast_1d(a.entropy()) == np.atleast_1d(b.entropy())).all()
    assert all((x == y).all() for x, y in zip(a.last_counts(), b.last_counts()))


@pytest.mark.parametrize("mode", ["philox", "compat"])
@pytest.mark.parametrize("graph,devices", [("n_1000", [0, 0]), ("southernWomen", [0, 0, 0])])
def test_devices_behind_one_handle_equal_one_device(graph, devices, mode):
    rowptr, col, na, nb = O.load_graph(graph)
    n = na + nb
    ka, kb, eps = (4, 6, 1.0) if graph == "n_1000" else (5, 5, 0.001)
    labels = O.contiguous_labels(na, nb, ka, kb)
    chains, first = 7, 3
    kw = dict(n_chains=chains, rng=mode, seed=91, gen_seed=17, first_chain_id=first)
    one = _model(rowptr, col, na, nb, ka, kb, eps, labels, device=0, **kw)
    many = _model(rowptr, col, na, nb, ka, kb, eps, labels, devices=devices, **kw)
    devs, firsts = many.device_layout()
    assert devs == devices and firsts == ([0, 4] if len(devices) == 2 else [0, 3, 5])  # 7 chains: 4 + 3, 3 + 2 + 2
    assert one.device_layout() == ([0], [0])
    mh = B.MetropolisHasting()
    for g in (one, many):
        g.shuffle_bisbm()
    _same_state(one, many, chains)
    for sched, kwargs, dur, aw in [("constant", [1.0], 5 * n, BIG), ("exponential", [2.0, 0.9995], 12 * n, 2 * n),
                                   ("abrupt_cool", [1.5 * n], 3 * n, BIG)]:
        ra, rb = mh.anneal(one, sched, kwargs, dur, aw).copy(), mh.anneal(many, sched, kwargs, dur, aw).copy()
        assert (ra == rb).all(), sched
        _same_state(one, many, chains)
    # ... and the oracle, chain by chain (global ids 3 .. 9)
    for c in (0, 3, 4, 6):
        o = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
        if mode == "compat":
            o.seed_compat(91 + first + c, 17 + first + c)
        else:
            o.seed_philox(91, first + c)
        o.shuffle_bisbm()
        o.anneal("constant", [1.0], 5 * n, BIG)
        o.anneal("exponential", [2.0, 0.9995], 12 * n, 2 * n)
        o.anneal("abrupt_cool", [1.5 * n], 3 * n, BIG)
        assert (o.memberships() == many.get_memberships(c)).all() and (o.m() == many.get_m(c)).all()
    # marginals: every device entry accumulates its chains, the histogram is pooled by node range on the device(s)
    for g in (one, many):
        g.marginals_reset()
        for _ in range(3):
            g.run_sweeps(1)
            g.marginals_accumulate(None)
    counts = one.marginals_get()
    assert counts.sum() == 3 * chains * n and (many.marginals_get() == counts).all()
    want = counts.argmax(axis=1) + np.where(np.arange(n) >= na, ka, 0)
    assert (one.marginals_map() == want).all() and (many.marginals_map() == want).all()
    with pytest.raises(B.BisbmError):  # a caller-owned device buffer belongs to one device
        many.marginals_accumulate(12345)
    # a per-chain write lands on the owner of the chain
    lab = one.get_memberships(5).copy()
    for g in (one, many):
        g.set_memberships(lab, chain=1)
        g.init_bisbm()
    _same_state(one, many, chains)
    # merges: the same change of counts everywhere, then every chain to its own end (shapes may differ per chain, per device)
    if graph == "n_1000":
        for g in (one, many):
            g.agg_merge(1, 2, 10)
        assert one.ka_kb(0) == many.ka_kb(6) == (3, 4)
        _same_state(one, many, chains)
        for g in (one, many):
            g.agg_merge(2, None, 10)
            mh.anneal(g, "abrupt_cool", [0.0], n, BIG)
        _same_state(one, many, chains)
        assert many.mixed_shapes == one.mixed_shapes
    one.close()
    many.close()


def _pooled(rowptr, col, na, nb, ka, kb, labels, chains, capfd, **kw):
    """shuffle, three sweeps with a sample of every chain each, a split of each type (max(KA, KB) grows: the pool's slices are
    allocated afresh), three more samples: MAP labels after both stages, the full histogram, and what BISBM_POOL_LOG said"""
    g = _model(rowptr, col, na, nb, ka, kb, 1.0, labels, n_chains=chains, rng="philox", seed=77, **kw)
    g.shuffle_bisbm()
    out = []
    capfd.readouterr()
    for stage in range(2):
        g.marginals_reset()
        for _ in range(3):
            g.run_sweeps(1)
            g.marginals_accumulate(None)
        out.append((g.marginals_map().copy(), g.marginals_get().copy()))
        if stage == 0:
            g.agg_merge(-1, -1, 4)
    log = capfd.readouterr().err
    g.close()
    return out, log


def test_rccl_branch_runs_with_one_rank(monkeypatch, capfd):
    """A multi-device handle over ONE distinct device takes the RCCL branch of the pooling (bisbm_multi.hip: librccl.so resolved
    with dlopen / dlsym, ncclCommInitAll, the grouped ncclReduceScatter, the argmax kernel, the grouped ncclAllGather) -- what a
    one-GPU box can execute of the exchange every multi-GPU node runs by default.  Its MAP labels must be the single-engine
    handle's and the peer-copy path's (BISBM_POOL=p2p), before and after max(KA, KB) has grown."""
    import torch
    rowptr, col, na, nb = O.load_graph("n_1000")
    n = na + nb
    labels = O.contiguous_labels(na, nb, 4, 6)
    monkeypatch.setenv("BISBM_POOL_LOG", "1")
    monkeypatch.delenv("BISBM_POOL", raising=False)
    before = torch.cuda.current_device()
    plain, log_plain = _pooled(rowptr, col, na, nb, 4, 6, labels, 9, capfd, device=0)
    rccl, log_rccl = _pooled(rowptr, col, na, nb, 4, 6, labels, 9, capfd, devices=[0])
    assert "[bisbm pool]" not in log_plain
    assert log_rccl.count("[bisbm pool] RCCL path: 1 communicator(s)") == 1 and "peer-copy" not in log_rccl, log_rccl
    monkeypatch.setenv("BISBM_POOL", "p2p")
    p2p, log_p2p = _pooled(rowptr, col, na, nb, 4, 6, labels, 9, capfd, devices=[0])
    assert "peer-copy path (BISBM_POOL=p2p)" in log_p2p and "RCCL path" not in log_p2p
    assert torch.cuda.current_device() == before
    for stage, (ka, kb) in enumerate(((4, 6), (5, 7))):
        counts = plain[stage][1]
        assert counts.shape == (n, max(ka, kb)) and counts.sum() == 3 * 9 * n
        want = counts.argmax(axis=1) + np.where(np.arange(n) >= na, ka, 0)
        for got in (plain, rccl, p2p):
            assert (got[stage][1] == counts).all()
            assert (got[stage][0] == want).all()


def _n_devices():
    import torch
    return torch.cuda.device_count()  # (counts devices without bringing the GPU up)


@pytest.mark.skipif(_n_devices() < 2, reason="needs two distinct GPUs")
@pytest.mark.parametrize("pool", ["rccl", "p2p"])
def test_two_distinct_devices_pool_like_one(pool, monkeypatch, capfd):
    """Two different ordinals behind one handle: the exchange proper -- RCCL between two ranks of this process, or peer copies
    with peer access enabled over the xGMI link -- against a single-device handle with all the chains.  (Skipped on a one-GPU
    box; there test_rccl_branch_runs_with_one_rank and the device-listed-twice rehearsal cover what can run.)"""
    import torch
    rowptr, col, na, nb = O.load_graph("n_1000")
    n = na + nb
    labels = O.contiguous_labels(na, nb, 4, 6)
    monkeypatch.setenv("BISBM_POOL_LOG", "1")
    if pool == "p2p":
        monkeypatch.setenv("BISBM_POOL", "p2p")
    else:
        monkeypatch.delenv("BISBM_POOL", raising=False)
    before = torch.cuda.current_device()
    plain, _ = _pooled(rowptr, col, na, nb, 4, 6, labels, 9, capfd, device=0)
    two, log = _pooled(rowptr, col, na, nb, 4, 6, labels, 9, capfd, devices=[0, 1])
    assert ("RCCL path: 2 communicator(s)" in log) if pool == "rccl" else ("peer-copy path (BISBM_POOL=p2p)" in log), log
    assert torch.cuda.current_device() == before  # the pooling calls visit every device and put the caller's back
    for stage in range(2):
        assert (two[stage][1] == plain[stage][1]).all() and (two[stage][0] == plain[stage][0]).all()


def test_cli_devices_flag_prints_what_one_device_prints():
    """`mcmc --devices 0,0 --chains N` == `mcmc --device 0 --chains N`, byte for byte on stdout: the annealing driver (the chain
    with the lowest description length is printed) and --marginalize (the histogram pooled over the devices)."""
    cli = os.path.join(ROOT, "bipartitesbm-mcmc_amd", "bin", "mcmc")
    el = os.path.join(O.GOLDEN, "bisbm-n_1000-ka_4-kb_6.edgelist")
    sizes = [125] * 4 + [84, 84, 83, 83, 83, 83]
    base = [cli, "-e", el, "-y", "500", "500", "-n", *map(str, sizes), "-z", "4", "6", "-E", "1", "-d", "5", "--rng", "philox",
            "--chains", "6", "--randomize"]
    for extra in (["-c", "exponential", "-a", "2", "0.9999", "-t", "30000", "-x", "4000"],
                  ["-b", "3000", "-t", "12000", "-f", "2000", "--marginalize"]):
        a = subprocess.run(base + extra + ["--device", "0"], capture_output=True, text=True)
        b = subprocess.run(base + extra + ["--devices", "0,0"], capture_output=True, text=True)
        c = subprocess.run(base + extra + ["--devices", "0,0,0,0"], capture_output=True, text=True)
        assert a.returncode == 0 and b.returncode == 0 and c.returncode == 0, (a.stderr, b.stderr, c.stderr)
        assert len(a.stdout.split()) == 1000 and a.stdout == b.stdout == c.stdout
        assert a.stderr == b.stderr  # (acceptance ratio, summary, which chain was printed)
    bad = subprocess.run(base + ["--devices", "0,x"], capture_output=True, text=True)
    assert bad.returncode == 1 and "Invalid --devices" in bad.stderr
    bad = subprocess.run(base + ["--devices", "0,0,0,0,0,0,0"], capture_output=True, text=True)
    assert bad.returncode == 1 and "every device needs a chain" in bad.stderr
