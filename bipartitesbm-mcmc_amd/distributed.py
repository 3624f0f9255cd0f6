"""Chain sharding over the GPUs of one node (one process per GPU, torch.distributed; backend
"nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The reference is single-chain and single-process (no parallel axis exists); the
engine adds exactly one: independent chains.  Chains never exchange anything during sweeps, so the
sweep kernels run with no collective.  Collectives appear only where chains are pooled:

  * all_gather of per-chain scalars (sum dS / description length, acceptance rate, counts);
  * marginals: every rank histograms its own chains into counts[n, kmax]; the pooled histogram is a
    reduce_scatter over node ranges, the MAP label an argmax on each rank's node range, the full
    label vector an all_gather of uint8 labels (int32 above 256 blocks).  On the fully connected xGMI topology
    a reduce_scatter moves 1/world of the buffer per link concurrently instead of a ring's
    per-link-bound all_reduce.

A chain's random stream is keyed by its GLOBAL chain id, so results do not depend on world_size.
"""
import numpy as np


def shard_chains(total_chains, world_size, rank):
    """Contiguous chain range of `rank`: returns (first_chain_id, n_local).  The first
    total % world ranks get one extra chain."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError("bad rank/world_size")
    base, extra = divmod(int(total_chains), int(world_size))
    n_local = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, n_local


def _dist():
    import torch.distributed as dist
    return dist


class ChainShard:
    """The chains one rank owns, plus the pooling collectives.

    `group` is a torch.distributed process group (None = default).  All tensors passed in must
    live on the device the backend needs (CUDA/HIP for nccl, CPU for gloo)."""

    def __init__(self, total_chains, rank=None, world_size=None, group=None):
        dist = _dist()
        self.group = group
        if world_size is None:
            world_size = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rank, self.world_size = int(rank), int(world_size)
        self.total_chains = int(total_chains)
        self.first_chain_id, self.n_local = shard_chains(total_chains, world_size, rank)
        self.counts = [shard_chains(total_chains, world_size, r)[1] for r in range(world_size)]

    # -- per-chain scalars ------------------------------------------------------------------
    def all_gather_chain_values(self, local):
        """local: tensor [n_local, ...] -> tensor [total_chains, ...] in global chain order."""
        import torch
        dist = _dist()
        if self.world_size == 1:
            return local.clone()
        pad = max(self.counts)
        buf = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        buf[: self.n_local] = local
        out = [torch.empty_like(buf) for _ in range(self.world_size)]
        dist.all_gather(out, buf, group=self.group)
        return torch.cat([o[:c] for o, c in zip(out, self.counts)], dim=0)

    # -- marginals --------------------------------------------------------------------------
    def node_range(self, n, rank=None):
        """Node rows [lo, hi) of the pooled histogram that `rank` reduces in map_labels (the n % world last rows are
        summed on every rank)."""
        rank = self.rank if rank is None else rank
        per = n // self.world_size
        return rank * per, (rank + 1) * per

    def pooled_marginals(self, local_counts):
        """all_reduce(sum) of counts[n, kmax] (int32): every rank gets the pooled histogram."""
        dist = _dist()
        out = local_counts.clone()
        if self.world_size > 1:
            dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def map_labels(self, local_counts, na, ka):
        """MAP block of every node from the pooled histogram: reduce_scatter by node range, argmax on
        the owned rows (ties -> lowest block, like numpy), all_gather of the labels.  Returns a tensor [n] of
        block indices in the reference's numbering (type-b blocks offset by ka): uint8 while every label fits a
        byte (ka + kmax <= 256), int32 otherwise (wide handles, two-byte labels inside the library).

        The histogram is handed to reduce_scatter_tensor as it is -- rows [0, world * (n // world)) are a view, not
        a padded copy (the buffer is 1 GB at BASELINE configs[4]); the fewer than `world` rows that do not divide
        evenly are summed with one small all_reduce and labelled on every rank."""
        import torch
        dist = _dist()
        n, kmax = local_counts.shape
        lab_dtype = torch.uint8 if int(ka) + int(kmax) <= 256 else torch.int32

        def label_rows(rows, first_node):
            node = torch.arange(first_node, first_node + rows.shape[0], device=rows.device)
            return (_argmax_first(rows) + torch.where(node >= na, ka, 0)).to(lab_dtype)

        if self.world_size == 1:
            return label_rows(local_counts, 0)
        per = n // self.world_size
        n_main = per * self.world_size
        parts = []
        if per > 0:
            mine = torch.empty((per, kmax), dtype=local_counts.dtype, device=local_counts.device)
            dist.reduce_scatter_tensor(mine, local_counts[:n_main], op=dist.ReduceOp.SUM, group=self.group)
            out = torch.empty(n_main, dtype=lab_dtype, device=mine.device)
            dist.all_gather_into_tensor(out, label_rows(mine, self.rank * per).contiguous(), group=self.group)
            parts.append(out)
        if n_main < n:
            tail = local_counts[n_main:].clone()
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
            parts.append(label_rows(tail, n_main))
        return torch.cat(parts) if len(parts) > 1 else parts[0]


def _argmax_first(counts):
    """argmax along dim 1 returning the FIRST maximal column (torch.argmax does not promise which)."""
    import torch
    mx = counts.max(dim=1, keepdim=True).values
    k = counts.shape[1]
    idx = torch.arange(k, device=counts.device).expand_as(counts)
    return torch.where(counts == mx, idx, torch.full_like(idx, k)).min(dim=1).values


def numpy_marginals(labels_by_chain, na, ka, kb):
    """Host restatement of the marginal histogram for small cases: labels_by_chain [chains, n] ->
    counts[n, max(ka,kb)] (used by tests to check the device kernel and the collectives)."""
    labels_by_chain = np.asarray(labels_by_chain)
    n = labels_by_chain.shape[1]
    kmax = max(ka, kb)
    counts = np.zeros((n, kmax), dtype=np.int32)
    base = np.where(np.arange(n) >= na, ka, 0)
    for row in labels_by_chain:
        np.add.at(counts, (np.arange(n), row.astype(np.int64) - base), 1)
    return counts
