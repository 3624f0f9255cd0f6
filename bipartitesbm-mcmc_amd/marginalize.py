"""Marginalisation driver: what the reference's README describes for "marginalize" mode
(`/root/reference/README.md:49-53,96-104`) and its CLI never implements (`-b/--burn_in` and
`-f/--sampling_frequency` are parsed at `src/mcmc_main.cc:61-65` and then unused).

Semantics (the README prose is the only specification): constant temperature T = 1; `burn_in` sweeps are
discarded; then `n_samples` samples are taken `sampling_frequency` sweeps apart; a sample adds every chain's
label of every node to a per-node histogram; the marginal estimate of a node is its most frequent block
(ties -> lowest block index).

Pooling over ranks stays on the device: every rank's chains are histogrammed by the marginals kernel
straight into a torch tensor on the rank's GPU (`bisbm_marginals_accumulate(device_counts)`), that tensor is
reduce-scattered by node range (`ChainShard.map_labels`: RCCL over xGMI with the nccl backend), each rank takes the
argmax of its rows, and the uint8 labels are all-gathered -- no host copy of the n x kmax histogram anywhere.  (With
the gloo backend of the CPU tests the same tensor is moved to the host first: gloo reduces host tensors.)
"""
import numpy as np


def marginalize(model, burn_in_sweeps, n_samples, sampling_frequency_sweeps, shard=None, device_counts=None,
                return_counts=None):
    """Runs the chain(s) of `model` (a BlockModel whose state is already initialised by init_bisbm() /
    shuffle_bisbm()) and returns (labels, counts):
      labels  uint32 [n]         MAP block of every node in the reference's numbering
      counts  [n, max(KA,KB)]    pooled histogram (column = block index within the node's type); None when
                                 return_counts is False.  Default: returned on a single rank, NOT returned when the chains
                                 are spread over ranks (there the pooled histogram costs an all_reduce of the whole
                                 n x kmax buffer, 1 GB at BASELINE configs[4], on top of the reduce_scatter the labels need)
    `shard`: a distributed.ChainShard when chains are spread over ranks.
    `device_counts`: a torch int32 tensor [n, kmax] on the model's device to accumulate into (it is NOT zeroed: samples
    add to what it holds); by default one is allocated when pooling over ranks, and the library's internal buffer is
    used for a single rank.  The library's kernels run on the handle's own (non-blocking) stream, so whatever torch
    still has in flight on the tensor (its zero fill, a caller's writes) is waited for here before the first sample
    is added -- the stream contract include/bisbm.h states for `device_counts`."""
    n = model.n
    multi = shard is not None and shard.world_size > 1
    if return_counts is None:
        return_counts = not multi
    if burn_in_sweeps > 0:
        model.run_sweeps(burn_in_sweeps)
    if device_counts is None and not multi:
        # one rank, no caller buffer: the library's own histogram
        model.marginals_reset()
        for _ in range(int(n_samples)):
            if sampling_frequency_sweeps > 0:
                model.run_sweeps(sampling_frequency_sweeps)
            model.marginals_accumulate(None)
        counts = model.marginals_get().astype(np.int64)
        base = np.where(np.arange(n) >= model.na, model.KA, 0)
        return (counts.argmax(axis=1) + base).astype(np.uint32), (counts if return_counts else None)

    import torch
    if device_counts is None:
        device_counts = torch.zeros((n, model.kmax), dtype=torch.int32, device=model.counts_device())
    if (not isinstance(device_counts, torch.Tensor) or device_counts.dtype != torch.int32
            or tuple(device_counts.shape) != (n, model.kmax) or not device_counts.is_contiguous()):
        raise ValueError("device_counts must be a contiguous torch.int32 tensor of shape (n, max(KA, KB)) on the model's device")
    if device_counts.is_cuda:
        # torch.zeros / the caller's kernels ran on torch's stream, bisbm_marginals_accumulate adds with plain (non-atomic)
        # read-modify-writes on the library's stream: order the two before the first sample.  (The other direction needs
        # nothing: bisbm_marginals_accumulate returns after its kernel has finished.)
        torch.cuda.current_stream(device_counts.device).synchronize()
    for _ in range(int(n_samples)):
        if sampling_frequency_sweeps > 0:
            model.run_sweeps(sampling_frequency_sweeps)
        model.marginals_accumulate(device_counts.data_ptr())  # adds into the tensor, on the device
    if not multi:
        from .distributed import _argmax_first
        arg = _argmax_first(device_counts)
        node = torch.arange(n, device=device_counts.device)
        labels = (arg + torch.where(node >= model.na, model.KA, 0)).cpu().numpy().astype(np.uint32)
        return labels, (device_counts.cpu().numpy().astype(np.int64) if return_counts else None)
    send = device_counts if _uses_cuda_backend(shard) else device_counts.cpu()
    labels = shard.map_labels(send, model.na, model.KA).cpu().numpy().astype(np.uint32)
    counts = shard.pooled_marginals(send).cpu().numpy().astype(np.int64) if return_counts else None
    return labels, counts


def _uses_cuda_backend(shard):
    import torch.distributed as dist
    return dist.is_initialized() and dist.get_backend(shard.group) == "nccl"
