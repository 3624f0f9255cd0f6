"""World-size-2 gloo tests (CPU) of the chain sharding and pooling layer
(bipartitesbm-mcmc_amd/distributed.py).  Per-rank chain results come from the oracle in Philox mode,
whose streams are keyed by the GLOBAL chain id -- the same property the GPU path relies on -- so the
pooled result must be independent of world_size."""
import importlib
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as O

B = importlib.import_module("bipartitesbm-mcmc_amd")
D = B.distributed

TOTAL_CHAINS = 7  # odd on purpose: uneven shards
SWEEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _run_chain(gid):
    rowptr, col, na, nb = O.load_graph("southernWomen")
    m = O.OracleModel(rowptr, col, na, nb, 5, 5, 0.001, O.contiguous_labels(na, nb, 5, 5))
    m.seed_philox(31337, gid)
    m.shuffle_bisbm()
    rate = m.anneal("constant", [1.0], SWEEPS * 32, 1 << 60)
    return m.memberships(), rate, m.get_entropy()


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = D.ChainShard(TOTAL_CHAINS)
        assert (shard.rank, shard.world_size) == (rank, world)
        labs, rates, cums = [], [], []
        for c in range(shard.n_local):
            l, r, e = _run_chain(shard.first_chain_id + c)
            labs.append(l)
            rates.append(r)
            cums.append(e)
        local = torch.tensor(np.stack([rates, cums], axis=1), dtype=torch.float64).reshape(-1, 2)
        allv = shard.all_gather_chain_values(local)
        counts = torch.from_numpy(D.numpy_marginals(np.array(labs).reshape(-1, 32), 18, 5, 5))
        pooled = shard.pooled_marginals(counts)
        mapl = shard.map_labels(counts, 18, 5)
        if rank == 0:
            np.save(os.path.join(out_dir, "allv.npy"), allv.numpy())
            np.save(os.path.join(out_dir, "pooled.npy"), pooled.numpy())
            np.save(os.path.join(out_dir, "map.npy"), mapl.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_shard_chains_partition():
    for total in (1, 7, 8, 1024, 8192):
        for world in (1, 2, 3, 8):
            spans = [D.shard_chains(total, world, r) for r in range(world)]
            assert spans[0][0] == 0
            assert sum(n for _, n in spans) == total
            for (f0, n0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + n0 == f1
            assert max(n for _, n in spans) - min(n for _, n in spans) <= 1
    with pytest.raises(ValueError):
        D.shard_chains(4, 2, 2)


def test_world2_gloo_pooling_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    allv = np.load(tmp_path / "allv.npy")
    pooled = np.load(tmp_path / "pooled.npy")
    mapl = np.load(tmp_path / "map.npy")
    # single-process truth
    labs, vals = [], []
    for gid in range(TOTAL_CHAINS):
        l, r, e = _run_chain(gid)
        labs.append(l)
        vals.append((r, e))
    assert allv.shape == (TOTAL_CHAINS, 2)
    assert (allv == np.array(vals)).all()  # global chain order, bit-exact
    want = D.numpy_marginals(np.array(labs), 18, 5, 5)
    assert (pooled == want).all()
    assert pooled.sum() == TOTAL_CHAINS * 32
    base = np.where(np.arange(32) >= 18, 5, 0)
    assert (mapl == want.argmax(axis=1) + base).all()


def _wide_counts(rank, n, na, ka, kb):
    rs = np.random.default_rng(100 + rank)
    c = rs.integers(0, 50, size=(n, max(ka, kb))).astype(np.int32)
    c[:na, ka:] = 0  # (columns past a type's block count stay empty, as the kernel leaves them)
    c[na:, kb:] = 0
    c[n - 1, kb - 1] = 10_000 // (rank + 1)  # the last node's MAP block is the very last block
    return c


def _wide_worker(rank, world, port, out_dir, n, na, ka, kb):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = D.ChainShard(4)
        lab = shard.map_labels(torch.from_numpy(_wide_counts(rank, n, na, ka, kb)), na, ka)
        assert lab.dtype == (torch.int32 if ka + max(ka, kb) > 256 else torch.uint8)
        if rank == 0:
            np.save(os.path.join(out_dir, "wide_map.npy"), lab.numpy())
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,na,ka,kb", [(33, 16, 300, 310), (32, 16, 300, 310), (33, 16, 100, 120), (3, 1, 2, 2)])
def test_world2_map_labels_above_256_blocks_and_uneven_rows(tmp_path, n, na, ka, kb):
    """Labels of a wide handle (two-byte labels inside the library) do not fit a byte: the pooled MAP labels come back as
    int32, and a node count that does not divide by the world size (the tail rows) is handled without a padded copy."""
    mp.spawn(_wide_worker, args=(2, _free_port(), str(tmp_path), n, na, ka, kb), nprocs=2, join=True)
    got = np.load(tmp_path / "wide_map.npy")
    pooled = _wide_counts(0, n, na, ka, kb).astype(np.int64) + _wide_counts(1, n, na, ka, kb)
    want = pooled.argmax(axis=1) + np.where(np.arange(n) >= na, ka, 0)
    assert got.shape == (n,) and (got == want).all()
    assert got[-1] == ka + kb - 1


def test_single_process_paths():
    shard = D.ChainShard(5, rank=0, world_size=1)
    x = torch.arange(10, dtype=torch.float64).reshape(5, 2)
    assert torch.equal(shard.all_gather_chain_values(x), x)
    counts = torch.tensor([[1, 3, 3], [2, 0, 1], [0, 0, 4]], dtype=torch.int32)
    assert torch.equal(shard.pooled_marginals(counts), counts)
    assert shard.map_labels(counts, 2, 3).tolist() == [1, 0, 5]


# ------------------------------------------------------------------ marginalize() end to end over two ranks
class _OracleChains:
    """Stands in for BlockModel in marginalize(): the rank's chains are oracle chains (Philox streams keyed by the global
    chain id, like the GPU's), the histogram buffer is the caller's torch tensor, written through its data pointer
    exactly as the HIP marginals kernel writes a device tensor."""

    def __init__(self, first_chain, n_local):
        rowptr, col, na, nb = O.load_graph("southernWomen")
        self.n, self.na, self.KA, self.KB = na + nb, na, 5, 5
        self.kmax = 5
        self.chains = []
        for c in range(n_local):
            m = O.OracleModel(rowptr, col, na, nb, 5, 5, 0.001, O.contiguous_labels(na, nb, 5, 5))
            m.seed_philox(31337, first_chain + c)
            m.shuffle_bisbm()
            self.chains.append(m)

    def run_sweeps(self, sweeps, temperature=1.0):
        for m in self.chains:
            m.anneal("constant", [temperature], sweeps * self.n, 1 << 60)

    def counts_device(self):
        return torch.device("cpu")

    def marginals_accumulate(self, ptr):
        import ctypes
        buf = np.ctypeslib.as_array((ctypes.c_int32 * (self.n * self.kmax)).from_address(ptr)).reshape(self.n, self.kmax)
        buf += D.numpy_marginals(np.array([m.memberships() for m in self.chains]).reshape(-1, self.n), self.na, 5, 5)

    def marginals_reset(self):
        raise AssertionError("the pooled path must not use the library's internal histogram")

    marginals_get = marginals_reset


def _marg_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shard = D.ChainShard(TOTAL_CHAINS)
        model = _OracleChains(shard.first_chain_id, shard.n_local)
        labels, counts = B.marginalize(model, 2, 3, 1, shard=shard, return_counts=True)
        # over several ranks the pooled histogram is not returned unless asked for (it costs an all_reduce of n x kmax)
        model2 = _OracleChains(shard.first_chain_id, shard.n_local)
        labels2, none = B.marginalize(model2, 2, 3, 1, shard=shard)
        assert none is None and (labels2 == labels).all()
        if rank == 0:
            np.save(os.path.join(out_dir, "m_labels.npy"), labels)
            np.save(os.path.join(out_dir, "m_counts.npy"), counts)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_world2_marginalize_end_to_end(tmp_path):
    port = _free_port()
    mp.spawn(_marg_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    labels = np.load(tmp_path / "m_labels.npy")
    counts = np.load(tmp_path / "m_counts.npy")
    # single process, all chains
    solo = _OracleChains(0, TOTAL_CHAINS)
    want_labels, want_counts = B.marginalize(solo, 2, 3, 1, device_counts=torch.zeros((32, 5), dtype=torch.int32))
    assert (counts == want_counts).all() and counts.sum() == TOTAL_CHAINS * 32 * 3
    assert (labels == want_labels).all()
    base = np.where(np.arange(32) >= 18, 5, 0)
    assert (labels == want_counts.argmax(axis=1) + base).all()
    with pytest.raises(ValueError):  # a raw pointer or a wrongly shaped buffer is refused, not silently ignored
        B.marginalize(solo, 0, 1, 1, device_counts=12345)
