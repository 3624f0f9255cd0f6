"""Marginalisation driver: what the reference's README describes for "marginalize" mode
(`/root/reference/README.md:49-53,96-104`) and its CLI never implements (`-b/--burn_in` and
`-f/--sampling_frequency` are parsed at `src/mcmc_main.cc:61-65` and then unused).

Semantics (the README prose is the only specification): constant temperature T = 1; `burn_in` sweeps are
discarded; then `n_samples` samples are taken `sampling_frequency` sweeps apart; a sample adds every chain's
label of every node to a per-node histogram; the marginal estimate of a node is its most frequent block
(ties -> lowest block index).  With several ranks the histograms are pooled with the collectives of
`distributed.ChainShard`.
"""
import numpy as np


def marginalize(model, burn_in_sweeps, n_samples, sampling_frequency_sweeps, shard=None, device_counts=None):
    """Runs the chain(s) of `model` (a BlockModel whose state is already initialised by init_bisbm() /
    shuffle_bisbm()) and returns (labels, counts):
      labels  uint8/uint32 [n]  MAP block of every node in the reference's numbering
      counts  [n, max(KA,KB)]   pooled histogram (column = block index within the node's type)
    `shard`: a distributed.ChainShard when chains are spread over ranks (pooling by RCCL / gloo)."""
    from . import MetropolisHasting, constant_schedule
    mh = MetropolisHasting()
    n = model.n
    big = 1 << 60
    if burn_in_sweeps > 0:
        mh.anneal(model, constant_schedule, [1.0], burn_in_sweeps * n, big)
    model.marginals_reset()
    for _ in range(int(n_samples)):
        if sampling_frequency_sweeps > 0:
            mh.anneal(model, constant_schedule, [1.0], sampling_frequency_sweeps * n, big)
        model.marginals_accumulate(device_counts)
    counts = model.marginals_get().astype(np.int64)
    if shard is not None and shard.world_size > 1:
        import torch
        t = torch.from_numpy(counts.astype(np.int32))
        if device_counts is not None or _uses_cuda_backend(shard):
            t = t.cuda()
        pooled = shard.pooled_marginals(t)
        labels = shard.map_labels(t, model.na, model.KA).cpu().numpy()
        return labels, pooled.cpu().numpy().astype(np.int64)
    base = np.where(np.arange(n) >= model.na, model.KA, 0)
    return (counts.argmax(axis=1) + base).astype(np.uint32), counts


def _uses_cuda_backend(shard):
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend(shard.group) == "nccl"
