#!/usr/bin/env python3
"""bench.py -- node-label MH updates/s of the sweep engine on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (configs[2] of BASELINE.json, the configuration the metric is quoted on): synthetic planted
bipartite graph N_a = N_b = 5e5, E = 1e7, Ka = Kb = 32 (graph seed 1), 1024
independent chains PER GPU, constant T = 1, epsilon = 1, randomised start ("marginalize" regime).
One step = one sweep = one pass of the hot path (n node updates) over every chain = one sweep-kernel
launch.  Inputs (CSR, labels, tables) are resident in HBM before the timed region starts.
Chains shard over GPUs with no collective in the sweep path (scaling = "weak": 1024 chains per GPU);
the RCCL pooling of per-chain sums runs after the timed region.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: sweep_kernel, HBM-bound, measured
with HIP events on the launch stream) and `cpu_baseline` (the oracle restatement timed on one host
core on a bounded sample of the same workload; N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md chip table: 8.0 TB/s spec (6.29 TB/s measured copy)


def b_alg_per_update(n, n_edges, label_bytes=1):
    """8 (two row offsets) + (4 + L) * mean degree (neighbour ids + labels) + 2 L."""
    return 8.0 + (4.0 + label_bytes) * (2.0 * n_edges / n) + 2.0 * label_bytes


def cpu_baseline(rowptr, col, na, nb, ka, kb, eps, labels, seconds_budget=20.0):
    """The oracle (kind "port": bit-exact with the reference in compat mode, see tests) timed on ONE
    host core: anneal() only, init excluded, constant T = 1 after a randomised start."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    n = na + nb
    m = O.OracleModel(rowptr, col, na, nb, ka, kb, eps, labels)
    m.seed_compat(42, 43)
    m.shuffle_bisbm()
    t0 = time.perf_counter()
    m.anneal("constant", [1.0], n, 1 << 60)  # one sweep to size the sample
    dt1 = time.perf_counter() - t0
    extra = int(max(0, min(50, (seconds_budget - dt1) // max(dt1, 1e-9))))
    steps, dt = n, dt1
    if extra > 0:
        t0 = time.perf_counter()
        m.anneal("constant", [1.0], extra * n, 1 << 60)
        dt += time.perf_counter() - t0
        steps += extra * n
    return {"value": steps / dt, "unit": "updates/s", "cores": 1, "kind": "port",
            "sample": "oracle/bisbm_oracle.c (mt19937-compat), 1 chain, %d sweeps of the same graph, "
                      "anneal() wall time only, %.1f s" % (steps // n, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spinup", type=int, default=2,
                    help="throw-away sweeps before the warm-up; up to 6 more while launch times still settle")
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU")
    ap.add_argument("--na", type=int, default=500_000)
    ap.add_argument("--nb", type=int, default=500_000)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--ka", type=int, default=32)
    ap.add_argument("--kb", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shuffle-ids", action="store_true",
                    help="custom workload: renumber the nodes of each type at random (ids that carry no structure)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # one process per GPU; BISBM_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed on a one-GPU box
    # (several ranks then share device 0 and the collectives run on CPU tensors)
    backend = os.environ.get("BISBM_BENCH_BACKEND", "nccl")
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    coll_device = torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("bipartitesbm-mcmc_amd")
    syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()

    na, nb, ka, kb, E = args.na, args.nb, args.ka, args.kb, args.edges
    n = na + nb
    a, b = syn.planted_edges(na, nb, E, ka, kb, seed=1)
    if args.shuffle_ids:
        import numpy as np
        rs = np.random.default_rng(7)
        pa, pb = rs.permutation(na).astype(a.dtype), rs.permutation(nb).astype(b.dtype)
        a, b = pa[a], pb[b - na] + na
    rowptr, col = pkg.edge_to_adj((a, b), n)
    del a, b
    labels = syn.contiguous_labels(na, nb, ka, kb)
    shard = pkg.ChainShard(args.chains * world, rank=rank, world_size=world)
    model = pkg.BlockModel(labels, syn.types_vector(na, nb), ka + kb, ka, kb, 1.0, (rowptr, col),
                           n_chains=shard.n_local, rng="philox", seed=20240229, device=device_index,
                           first_chain_id=shard.first_chain_id)
    model.shuffle_bisbm()  # --randomize start
    mh = pkg.MetropolisHasting()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Spin-up, not part of the protocol's W + K steps: the chains simply run on through a few throw-away sweeps -- at
    # least --spinup of them, then until two launches in a row are within 5 % of the fastest one seen -- so the warm-up
    # and timed sweeps are later sweeps of the same chains (slightly slower ones: the accepted fraction and with it
    # the speed drift down by ~0.5 % per sweep).  History: before the kernel assigned its stepping waves to SIMDs
    # itself, one launch in three or so came out 22 % slow (two stepping waves on one SIMD);
    # the settle test is the guard that remains from that.
    verbose = bool(os.environ.get("BISBM_BENCH_VERBOSE"))

    def note(what):
        if verbose:
            print("%s launch %.1f ms, accepted %.4f, ended at %.3f" % (
                what, model.last_sweep_timing()[0], float(model.last_counts()[0].sum()) / (n * shard.n_local),
                time.time()), file=sys.stderr, flush=True)

    spin_ms = []
    while args.spinup and len(spin_ms) < args.spinup + 6:
        mh.anneal(model, pkg.constant_schedule, [1.0], n, 1 << 60)
        note("spin-up")
        spin_ms.append(model.last_sweep_timing()[0])
        if len(spin_ms) >= max(args.spinup, 2) and max(spin_ms[-2:]) <= 1.05 * min(spin_ms):
            break
    for _ in range(args.warmup):
        mh.anneal(model, pkg.constant_schedule, [1.0], n, 1 << 60)
        note("warm-up")
    sync()
    t0 = time.perf_counter()
    kernel_ms, updates = 0.0, 0
    for _ in range(args.steps):
        mh.anneal(model, pkg.constant_schedule, [1.0], n, 1 << 60)  # blocks until the sweep kernel is done
        ms, upd = model.last_sweep_timing()
        note("timed")
        kernel_ms += ms
        updates += upd
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([updates], dtype=torch.int64, device=coll_device)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_updates = int(u.item())
    else:
        total_updates = updates

    # pooling epilogue (outside the timed region): RCCL all_gather of the per-chain sums
    cum = torch.from_numpy(model.get_entropy()).to(coll_device).reshape(-1, 1)
    t1 = time.perf_counter()
    allcum = shard.all_gather_chain_values(cum)
    torch.cuda.synchronize()
    pool_ms = (time.perf_counter() - t1) * 1e3
    assert allcum.shape[0] == args.chains * world and bool(torch.isfinite(allcum).all())

    if rank == 0:
        per_launch_updates = updates / max(args.steps, 1)
        avg_kernel_s = kernel_ms / max(args.steps, 1) / 1e3
        balg = b_alg_per_update(n, E)
        achieved = balg * per_launch_updates / avg_kernel_s / 1e9
        default_cfg = (na, nb, E, ka, kb, args.chains) == (500_000, 500_000, 10_000_000, 32, 32, 1024) and not args.shuffle_ids
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
        # WRITE_SIZE, separate runs of this same command); only quoted for the workload they were measured on
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if default_cfg and os.path.exists(tpath):
            with open(tpath) as f:
                tj = json.load(f)
            traffic = tj["bytes_per_update"] * per_launch_updates
            traffic_src = "profiles/r01_traffic.json (FETCH_SIZE+WRITE_SIZE, %.0f B per update)" % tj["bytes_per_update"]
        out = {
            "metric": "node-label MH updates/s",
            "value": total_updates / elapsed,
            "unit": "updates/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / max(args.steps, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "sweeps_per_s": total_updates / elapsed / n,
            "config": {
                "workload": ("BASELINE configs[2]: " if default_cfg else "custom: ")
                + "planted bipartite N_a=%d N_b=%d E=%d Ka=%d Kb=%d, %d chains/GPU, constant T=1, eps=1, "
                  "randomised start, Philox mode%s" % (na, nb, E, ka, kb, args.chains,
                                                       ", node ids renumbered at random" if args.shuffle_ids else ""),
                "chains_total": args.chains * world, "step": "one sweep (n node updates) of every chain",
                "spinup_sweeps_before_warmup": len(spin_ms),
                "parallelism": "chains sharded, no collective in the sweep path",
            },
            "roofline": {
                "bound": "hbm", "kernel": "sweep_fast_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "alg_bytes_per_update": balg, "updates_per_launch": per_launch_updates,
                "avg_launch_ms": avg_kernel_s * 1e3,
            },
            "pool_all_gather_ms": pool_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rowptr, col, na, nb, ka, kb, 1.0, labels)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
