#!/usr/bin/env python3
"""bench.py -- node-label MH updates/s of the sweep engine on BASELINE.json's headline workload.

  python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher in the environment (RANK / WORLD_SIZE unset) this process starts the N ranks itself
-- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...` as a
child, before anything here has touched the GPU -- forwards rank 0's JSON line and exits with the children's
code.  Under a launcher (RANK / WORLD_SIZE set) it is one rank.

Workload (configs[2] of BASELINE.json, the configuration the metric is quoted on): synthetic planted
bipartite graph N_a = N_b = 5e5, E = 1e7, Ka = Kb = 32 (graph seed 1), 1024
independent chains PER GPU, constant T = 1, epsilon = 1, randomised start ("marginalize" regime).
One step = one sweep = one pass of the hot path (n node updates) over every chain = one sweep-kernel
launch.  Inputs (CSR, labels, tables) are resident in HBM before the timed region starts.
Chains shard over GPUs with no collective in the sweep path (scaling = "weak": 1024 chains per GPU);
the RCCL pooling of per-chain sums and of the marginal histogram runs after the timed region.

Prints ONE JSON line on rank 0 with `roofline` (dominant kernel: sweep_fast_kernel; the HBM figure SURVEY 8(d)
defines, the instruction-issue ceiling the counters support, and the measured HBM traffic) and `cpu_baseline` (the
oracle restatement timed on the host cores on a bounded sample of the same workload; N = 1 only).
"""
import argparse
import importlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md chip table: 8.0 TB/s spec (6.29 TB/s measured copy)
N_SIMDS = 256 * 4      # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md chip table)
ISSUE_CYCLES = 4.0     # a wave64 VALU instruction occupies its SIMD for 4 cycles (same guide, cycle constants)


def build_tag():
    """What the numbers of a run belong to: a hash of the library's sources (the same on the build container and on the GPU
    box; the .so itself is rebuilt per machine).  Profile summaries under profiles/ carry the tag of the build they were
    measured on; bench.py quotes them as current only when it matches."""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "bipartitesbm-mcmc_amd", "csrc")
    for name in sorted(os.listdir(base)):
        if name.endswith((".hip", ".hpp", ".cpp")):
            with open(os.path.join(base, name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
    with open(os.path.join(ROOT, "bipartitesbm-mcmc_amd", "build.py"), "rb") as f:  # (the compiler flags are part of a build)
        h.update(b"build.py\0" + f.read())
    return h.hexdigest()[:12]


def b_alg_per_update(n, n_edges, label_bytes=1):
    """8 (two row offsets) + (4 + L) * mean degree (neighbour ids + labels) + 2 L."""
    return 8.0 + (4.0 + label_bytes) * (2.0 * n_edges / n) + 2.0 * label_bytes


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spinup", type=int, default=4,
                    help="throw-away sweeps before the warm-up (four: the library launches every pass depth twice before it "
                         "trusts a measurement); up to 6 more while launch times still settle")
    ap.add_argument("--chains", type=int, default=None, help="chains per GPU (default 1024; 256 for --edgelist n_1000)")
    ap.add_argument("--na", type=int, default=500_000)
    ap.add_argument("--nb", type=int, default=500_000)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--ka", type=int, default=32)
    ap.add_argument("--kb", type=int, default=32)
    ap.add_argument("--edgelist", default=None,
                    help="run on this edge-list file instead of the synthetic graph (--na / --nb give the type sizes, --ka / --kb the "
                         "blocks of the contiguous initial partition); --edgelist n_1000 = the reference's shipped 1000-node data "
                         "set as BASELINE configs[1] runs it (Ka = 4, Kb = 6, 256 chains, 2000 sweeps per step)")
    ap.add_argument("--steady-sweeps", type=int, default=None,
                    help="after the timed region let the chains run on until they have done this many sweeps in all, then time "
                         "three more: the steady-state figure, measured live.  Default: 150 on the default workload with one GPU "
                         "(~1.5 min; --no-extras skips it), 0 otherwise")
    ap.add_argument("--sweeps-per-step", type=int, default=1, help="sweeps of every chain per timed step (small graphs: one launch should last milliseconds)")
    ap.add_argument("--rng", choices=["philox", "compat"], default="philox",
                    help="compat: the mt19937-compat path (the reference's own random streams, draw order and summation order -- the "
                         "only path that is bit-exact with the reference; a verification mode, one generic-kernel wave per chain)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra legs after the timed region (equilibrated-start figure, pooling timings)")
    ap.add_argument("--shuffle-ids", action="store_true",
                    help="custom workload: renumber the nodes of each type at random (ids that carry no structure)")
    ap.add_argument("--planted-start", action="store_true",
                    help="custom workload: start every chain on the generator's planted partition instead of a randomised one")
    ap.add_argument("--no-reorder", action="store_true",
                    help="with --shuffle-ids: do not run the ingest-time locality reordering pass")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)  # internal: one CPU-baseline process
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_launch(args):
    """--gpus N > 1 outside a launcher: start the N ranks as children (never re-exec: this process has not touched the
    GPU and never will), forward rank 0's JSON line, return the children's exit code."""
    pkg = importlib.import_module("bipartitesbm-mcmc_amd")  # ctypes + numpy only: no HIP call
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()  # once, here, so the ranks do not race to compile it
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for out in proc.stdout:
        s = out.strip()
        if s.startswith("{") and '"metric"' in s:
            line = s
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
        sys.stderr.write("bench.py: the ranks exited without a result line\n")
    return rc


# ----------------------------------------------------------------------------------------------- workload
def apply_presets(args):
    """--edgelist n_1000: BASELINE configs[1] (the shipped data set, copied as a fixture under tests/golden/)."""
    if args.edgelist == "n_1000":
        args.edgelist = os.path.join(ROOT, "tests", "golden", "bisbm-n_1000-ka_4-kb_6.edgelist")
        args.preset = "BASELINE configs[1]"
        args.na, args.nb, args.ka, args.kb = 500, 500, 4, 6
        if args.chains is None:
            args.chains = 256
        if args.sweeps_per_step == 1:
            args.sweeps_per_step = 2000
    else:
        args.preset = None
    if args.chains is None:
        args.chains = 1024
    return args


def make_graph(args, pkg, syn):
    na, nb, ka, kb, E = args.na, args.nb, args.ka, args.kb, args.edges
    n = na + nb
    if args.edgelist:
        rowptr, col = pkg.load_graph(args.edgelist, n)
        args.edges = int(rowptr[-1]) // 2
        return rowptr, col, syn.contiguous_labels(na, nb, ka, kb)
    a, b = syn.planted_edges(na, nb, E, ka, kb, seed=1)
    planted = syn.contiguous_labels(na, nb, ka, kb)  # the generator's own partition
    if args.shuffle_ids:
        rs = np.random.default_rng(7)
        pa, pb = rs.permutation(na).astype(a.dtype), rs.permutation(nb).astype(b.dtype)
        a, b = pa[a], pb[b - na] + na
        moved = np.empty_like(planted)
        moved[np.concatenate([pa, pb + na]).astype(np.int64)] = planted
        planted = moved
    rowptr, col = pkg.edge_to_adj((a, b), n)
    return rowptr, col, planted


# ----------------------------------------------------------------------------------------------- CPU baseline
# The reference's own binary cannot be built on this image (Boost) -- SURVEY section 6 timed it once in the build
# container with a header shim: 2.27e5 updates/s on one core of an "Intel Xeon Processor @ 2.10GHz" at this workload
# (22.0 s per 5 sweeps), where the port (the oracle, same flags) runs 9.9e5 updates/s.
REF_PROBE = {"reference_updates_per_s": 2.27e5, "port_updates_per_s_same_cpu": 9.9e5,
             "cpu": "Intel Xeon Processor @ 2.10GHz (build container, 1 core)",
             "source": "SURVEY.md section 6 [probe] (reference TUs -O3, anneal() only, 5 sweeps) and DESIGN.md section 7"}
# ... and on the shipped n_1000 data set (Ka = 4, Kb = 6, T = 1): 3.4e6 updates/s; the port on the same CPU: 5.0e6
REF_PROBE_N1000 = {"reference_updates_per_s": 3.4e6, "port_updates_per_s_same_cpu": 5.0e6,
                   "cpu": "Intel Xeon Processor @ 2.10GHz (build container, 1 core)",
                   "source": "SURVEY.md section 6 [probe] (n_1000, 2e6 steps) and the port timed in the same container"}


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_worker(args):
    """One process of the CPU baseline: the oracle (kind "port": bit-exact with the reference's recorded runs in compat
    mode, see tests) on one core: anneal() only, init excluded, constant T = 1 after a randomised start."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
    args = apply_presets(args)
    na, nb, ka, kb = args.na, args.nb, args.ka, args.kb
    n = na + nb
    if args.edgelist:
        a, b = O.load_edge_list(args.edgelist)
    else:
        a, b = syn.planted_edges(na, nb, args.edges, ka, kb, seed=1)
    rowptr, col = O.edge_to_csr(a, b, n)
    del a, b
    m = O.OracleModel(rowptr, col, na, nb, ka, kb, 1.0, syn.contiguous_labels(na, nb, ka, kb))
    m.seed_compat(42 + int(os.environ.get("BISBM_CPU_WORKER_INDEX", "0")), 43)
    m.shuffle_bisbm()
    print("ready", flush=True)
    sys.stdin.readline()  # all workers start their timed part together
    budget = args.cpu_worker
    t0 = time.perf_counter()
    m.anneal("constant", [1.0], n, 1 << 60)  # one sweep to size the sample
    dt = time.perf_counter() - t0
    steps = n
    extra = int(max(0, min(max(50, 20_000_000 // n), (budget - dt) // max(dt, 1e-9))))
    if extra > 0:
        t0 = time.perf_counter()
        m.anneal("constant", [1.0], extra * n, 1 << 60)
        dt += time.perf_counter() - t0
        steps += extra * n
    rss = 0
    try:
        with open("/proc/self/status") as f:
            for ln in f:
                if ln.startswith("VmHWM"):
                    rss = int(ln.split()[1]) * 1024
    except OSError:
        pass
    print(json.dumps({"steps": steps, "seconds": dt, "rss": rss}), flush=True)


def _cpu_limits():
    """What this process may use of the host: logical CPUs of the box, CPUs in the affinity mask, the CPU quota of the
    control group it runs in (cgroup v2 cpu.max / v1 cfs quota; None: unlimited), and the memory still available."""
    total = os.cpu_count() or 1
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = total
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = int(f.read()), int(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    mem = None
    try:
        with open("/proc/meminfo") as f:
            for ln in f:
                if ln.startswith("MemAvailable"):
                    mem = int(ln.split()[1]) * 1024
        for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):
            if os.path.exists(path):
                with open(path) as f:
                    v = f.read().strip()
                if v != "max" and int(v) < (1 << 60):
                    mem = min(mem, int(v)) if mem else int(v)
                break
    except (OSError, ValueError):
        pass
    return total, affinity, quota, mem


def _worker_rss_bytes(n, n_edges):
    """Resident set of one oracle process at this workload (measured: 1.35 GB at N = 10^6, E = 10^7; the workers report their own, `rss_per_worker_bytes`): interpreter + numpy
    ~0.1 GB, the generator's edge arrays and the CSR, and the oracle's state (it keeps the reference's N x K neighbour-count
    matrix)."""
    return int(0.15e9 + 90.0 * n_edges + 300.0 * n)


def cpu_baseline(n, n_edges, seconds_budget=18.0, probe=None):
    """Independent single-chain processes of the oracle, one per host core this process may use, timed together (the reference
    is single-threaded: its multi-core figure is one process per core).  value = their summed rate.  Workers = the
    CPUs of the affinity mask, capped by the control group's CPU quota (more runnable processes than the quota only take turns)
    and by memory (every worker holds the graph and the oracle's state)."""
    total, affinity, quota, mem = _cpu_limits()
    cores = affinity
    if quota is not None:
        cores = min(cores, max(1, int(quota + 0.5)))
    rss = _worker_rss_bytes(n, n_edges)
    if mem:
        cores = min(cores, max(1, int(0.5 * mem // rss)))
    env_cap = os.environ.get("BISBM_BENCH_CPU_WORKERS")
    if env_cap:
        cores = min(cores, max(1, int(env_cap)))
    cores = max(1, cores)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", str(seconds_budget)] + \
          [a for a in sys.argv[1:] if a not in ("--no-extras",)]
    procs = [subprocess.Popen(cmd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True,
                              env=dict(os.environ, BISBM_CPU_WORKER_INDEX=str(i), OMP_NUM_THREADS="1"))
             for i in range(cores)]
    for p in procs:
        assert p.stdout.readline().strip() == "ready"
    for p in procs:
        p.stdin.write("go\n")
        p.stdin.flush()
    res = []
    for p in procs:
        res.append(json.loads(p.stdout.readline()))
        p.wait()
    per_core = [r["steps"] / r["seconds"] for r in res]
    total_rate = float(sum(per_core))
    one = float(np.median(per_core))
    pr = probe or REF_PROBE
    return {"value": total_rate, "unit": "updates/s", "cores": cores, "kind": "port",
            "cores_total": total, "cores_in_affinity_mask": affinity, "cgroup_cpu_quota": quota,
            "rss_per_worker_bytes": int(max(r.get("rss", 0) for r in res)), "memory_available_bytes": mem,
            "per_core": one, "cpu": _cpu_model(),
            "sample": "oracle/bisbm_oracle.c (mt19937-compat), %d independent 1-chain processes (one per usable core), %d-%d "
                      "sweeps each of the same graph, anneal() wall time only, %.1f s" % (
                          cores, min(r["steps"] for r in res) // n, max(r["steps"] for r in res) // n,
                          max(r["seconds"] for r in res)),
            # The reference's own binary cannot be built on this image (Boost).  What is known about it is one probe on ANOTHER
            # CPU (the build container's Xeon): there the port ran `ratio` times faster than the reference at this workload, whose
            # cost is cache misses into the reference's dense N x K matrix -- so the ratio need not carry over to this host.
            "reference_on_another_cpu": {"port_over_reference_there": pr["port_updates_per_s_same_cpu"] / pr["reference_updates_per_s"],
                                         "estimate_for_this_host_updates_per_s": total_rate * pr["reference_updates_per_s"] / pr["port_updates_per_s_same_cpu"],
                                         "caveat": "ratio measured on another CPU (smaller L3): an estimate, not a measurement on this box",
                                         "probe": pr}}


# ----------------------------------------------------------------------------------------------- main
def profile_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        with open(path) as f:
            return json.load(f)
    return None


def main():
    args = parse_args()
    if args.cpu_worker > 0:
        return cpu_worker(args)
    args = apply_presets(args)
    in_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not in_launcher:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    # one process per GPU; BISBM_BENCH_BACKEND=gloo lets the multi-rank path be rehearsed on a one-GPU box
    # (several ranks then share device 0 and the collectives run on CPU tensors)
    backend = os.environ.get("BISBM_BENCH_BACKEND", "nccl")
    backend_note = None
    device_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(device_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            try:
                import datetime
                dist.init_process_group("nccl", device_id=torch.device("cuda", device_index),
                                        timeout=datetime.timedelta(seconds=180))
                probe = torch.ones(1, device=torch.device("cuda", device_index))
                dist.all_reduce(probe)  # the first collective is where a broken RCCL setup shows
                torch.cuda.synchronize()
            except Exception as exc:  # the sweep path has no collective: the measurement does not depend on RCCL
                backend_note = "nccl unusable (%s: %s); timing coordination and pooling over gloo" % (type(exc).__name__, str(exc)[:200])
                print("bench.py rank %d: %s" % (rank, backend_note), file=sys.stderr, flush=True)
                try:
                    dist.destroy_process_group()
                except Exception:
                    pass
                # (a store of our own: under torchrun the workers are clients of the agent's store on MASTER_PORT)
                import datetime as _dt
                store = dist.TCPStore("127.0.0.1", int(os.environ.get("MASTER_PORT", "29500")) + 1, world, is_master=(rank == 0),
                                      timeout=_dt.timedelta(seconds=120))
                backend = "gloo"
                dist.init_process_group("gloo", store=store, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend)
    coll_device = torch.device("cuda", device_index) if backend == "nccl" else torch.device("cpu")

    pkg = importlib.import_module("bipartitesbm-mcmc_amd")
    syn = importlib.import_module("bipartitesbm-mcmc_amd.synthetic")
    if not os.path.exists(pkg.LIB_PATH):
        pkg.build()  # (under a file lock: with several ranks one compiles, the others wait)

    na, nb, ka, kb, E = args.na, args.nb, args.ka, args.kb, args.edges
    n = na + nb
    rowptr, col, planted = make_graph(args, pkg, syn)
    E = args.edges  # (an --edgelist run: the file's edge count)
    reorder = None
    if args.shuffle_ids and not args.no_reorder and hasattr(pkg, "locality_order"):
        # ingest-time reordering pass for ids that carry no structure: the engine runs on the
        # renumbered graph; labels are mapped back at the boundary
        t0 = time.perf_counter()
        reorder = pkg.locality_order(rowptr, col, na, nb)
        rowptr, col = reorder.apply(rowptr, col)
        planted = reorder.to_new(planted)
        reorder_s = time.perf_counter() - t0
    labels = syn.contiguous_labels(na, nb, ka, kb)
    shard = pkg.ChainShard(args.chains * world, rank=rank, world_size=world)
    model = pkg.BlockModel(labels, syn.types_vector(na, nb), ka + kb, ka, kb, 1.0, (rowptr, col),
                           n_chains=shard.n_local, rng=args.rng, seed=20240229, gen_seed=20240301, device=device_index,
                           first_chain_id=shard.first_chain_id)
    if args.planted_start:
        model.set_memberships(planted)
        model.init_bisbm()
    else:
        model.shuffle_bisbm()  # --randomize start
    mh = pkg.MetropolisHasting()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    # Spin-up, not part of the protocol's W + K steps: the chains simply run on through a few throw-away sweeps -- at
    # least --spinup of them, then until two launches in a row are within 5 % of the fastest one seen -- so the warm-up
    # and timed sweeps are later sweeps of the same chains (slightly slower ones: the accepted fraction and with it
    # the speed drift down by ~0.5 % per sweep).
    verbose = bool(os.environ.get("BISBM_BENCH_VERBOSE"))

    def note(what):
        if verbose:
            print("%s launch %.1f ms, accepted %.4f, ended at %.3f" % (
                what, model.last_sweep_timing()[0], float(model.last_counts()[0].sum()) / (n * args.sweeps_per_step * shard.n_local),
                time.time()), file=sys.stderr, flush=True)

    def sweep():  # one timed step: --sweeps-per-step sweeps of every chain in one launch
        mh.anneal(model, pkg.constant_schedule, [1.0], n * args.sweeps_per_step, 1 << 60)  # blocks until the sweep kernel is done

    spin_ms = []
    while args.spinup and len(spin_ms) < args.spinup + 6:
        sweep()
        note("spin-up")
        spin_ms.append(model.last_sweep_timing()[0])
        if len(spin_ms) >= max(args.spinup, 2) and max(spin_ms[-2:]) <= 1.05 * min(spin_ms):
            break
    for _ in range(args.warmup):
        sweep()
        note("warm-up")
    sync()
    t0 = time.perf_counter()
    kernel_ms, updates = 0.0, 0
    pass_steps = []  # steps per pass of each timed launch (chosen per launch by the library; the chain does not depend on it)
    launch_ms, launch_acc = [], []  # ... its kernel time and the fraction of its steps that were accepted
    for _ in range(args.steps):
        sweep()
        ms, upd = model.last_sweep_timing()
        pass_steps.append(model.last_pass_steps())
        launch_ms.append(round(ms, 2))
        launch_acc.append(round(float(model.last_counts()[0].sum()) / max(upd, 1), 4))
        note("timed")
        kernel_ms += ms
        updates += upd
    sync()
    elapsed = time.perf_counter() - t0
    accepted_frac = float(model.last_counts()[0].sum()) / (n * args.sweeps_per_step * shard.n_local)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        u = torch.tensor([updates], dtype=torch.int64, device=coll_device)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
        total_updates = int(u.item())
    else:
        total_updates = updates

    # ---- after the timed region -------------------------------------------------------------------------------------
    extras = {}
    # pooling epilogue: RCCL all_gather of the per-chain sums
    cum = torch.from_numpy(model.get_entropy()).to(coll_device).reshape(-1, 1)
    sync()
    t1 = time.perf_counter()
    allcum = shard.all_gather_chain_values(cum)
    torch.cuda.synchronize()
    pool_ms = (time.perf_counter() - t1) * 1e3
    assert allcum.shape[0] == args.chains * world and bool(torch.isfinite(allcum).all())
    if world > 1 and not args.no_extras:
        extras["pool_all_gather_ms"] = pool_ms
        # marginals: one sample of every chain into a device histogram, pooled by reduce_scatter over node ranges
        # (128 MB of counts at this workload, 1 GB at config 5's)
        dev = torch.device("cuda", device_index)
        counts = torch.zeros((n, model.kmax), dtype=torch.int32, device=dev)
        torch.cuda.current_stream(dev).synchronize()  # (the zero fill is torch's stream, the histogram kernel the library's)
        model.marginals_accumulate(counts.data_ptr())
        torch.cuda.synchronize()
        send = counts if backend == "nccl" else counts.cpu()
        sync()
        t1 = time.perf_counter()
        lab = shard.map_labels(send, na, ka)
        if backend == "nccl":
            torch.cuda.synchronize()
        extras["pool_marginals_reduce_scatter_argmax_all_gather_ms"] = (time.perf_counter() - t1) * 1e3
        extras["pool_marginals_bytes"] = int(counts.numel() * 4)
        assert lab.shape[0] == n
        del counts, send

    # the same chains further along (a marginalize run lives there, not in the burn-in the protocol's sweeps see)
    if args.steady_sweeps is None:
        is_default = ((na, nb, args.edges, ka, kb, args.chains) == (500_000, 500_000, 10_000_000, 32, 32, 1024) and not args.edgelist
                      and not args.shuffle_ids and not args.planted_start and args.rng == "philox" and args.sweeps_per_step == 1)
        args.steady_sweeps = 150 if (is_default and world == 1 and not args.no_extras) else 0
    steady_live = None
    sweeps_so_far = (len(spin_ms) + args.warmup + args.steps) * args.sweeps_per_step
    if args.steady_sweeps > sweeps_so_far:
        left = args.steady_sweeps - sweeps_so_far
        while left > 0:  # (long calls at constant T run as several launches inside the library anyway)
            now = min(left, 25)
            mh.anneal(model, pkg.constant_schedule, [1.0], n * now, 1 << 60)
            left -= now
        sync()
        sms, supd = 0.0, 0
        for _ in range(3):
            sweep()
            ms, upd = model.last_sweep_timing()
            sms += ms
            supd += upd
        steady_live = {"updates_per_s_per_gpu": supd / (sms / 1e3), "avg_launch_ms": sms / 3, "sweeps_before": args.steady_sweeps,
                       "accepted_fraction": float(model.last_counts()[0].sum()) / (n * args.sweeps_per_step * shard.n_local),
                       "what": "the same chains after %d sweeps in all, 3 sweeps timed (kernel time), measured in this run" % args.steady_sweeps}

    # the same workload from an equilibrated start: every chain on the planted partition (the posterior mode of this
    # generator), a few sweeps to settle, then timed -- the regime a long marginalize run lives in
    equil = None
    if not args.no_extras:
        model.set_memberships(planted)
        model.init_bisbm()
        for _ in range(3):
            sweep()
        sync()
        ems, eupd = 0.0, 0
        for _ in range(3):
            sweep()
            ms, upd = model.last_sweep_timing()
            ems += ms
            eupd += upd
        equil = {"updates_per_s_per_gpu": eupd / (ems / 1e3), "avg_launch_ms": ems / 3, "steps_per_pass": model.last_pass_steps(),
                 "accepted_fraction": float(model.last_counts()[0].sum()) / (n * args.sweeps_per_step * shard.n_local),
                 "what": "same graph and chains, started on the planted partition (near the posterior mode), 3 sweeps "
                         "to settle, 3 timed (kernel time)"}

    if rank == 0:
        per_launch_updates = updates / max(args.steps, 1)
        avg_kernel_s = kernel_ms / max(args.steps, 1) / 1e3
        balg = b_alg_per_update(n, E)
        achieved = balg * per_launch_updates / avg_kernel_s / 1e9
        default_cfg = ((na, nb, E, ka, kb, args.chains) == (500_000, 500_000, 10_000_000, 32, 32, 1024) and not args.shuffle_ids
                       and not args.planted_start and not args.edgelist and args.sweeps_per_step == 1 and args.rng == "philox")
        # HBM bytes per launch and instructions per update from the PMC passes committed under profiles/ (rocprofv3
        # --pmc, separate runs of this same command); only quoted for the workload they were measured on
        traffic, traffic_src, issue, steady = None, None, None, None
        tag = build_tag()

        def tagged(name):  # a profile summary and whether it was measured on THIS build
            j = profile_json(name)
            if j is None:
                return None, False
            return j, j.get("build") == tag

        if default_cfg:
            tj, cur = tagged("r04_traffic.json")
            if tj:
                traffic = tj["bytes_per_update"] * per_launch_updates
                traffic_src = "%s (FETCH_SIZE+WRITE_SIZE, %.0f B per update%s; measured on build %s%s)" % (
                    tj.get("_file", "profiles/r04_traffic.json"), tj["bytes_per_update"],
                    ", calibration: " + tj["calibration"] if "calibration" in tj else "", tj.get("build"),
                    "" if cur else " -- NOT this build (%s): re-take with tools/profile.sh" % tag)
            ij, cur = tagged("r04_issue.json")
            if ij:
                # Instruction-issue ceilings.  All kinds: a lone stepping wave per SIMD issues one instruction of any kind per
                # >= 4 cycles, so updates/s <= SIMDs x clock / (instructions per update x 4).  VALU only: what the vector
                # unit alone allows if scalar / LDS / memory instructions of other waves filled every other slot.
                ipu, clk, vpu = ij["instructions_per_update"], ij["clock_ghz"], ij["valu_per_update"]
                peak = N_SIMDS * clk * 1e9 / (ipu * ISSUE_CYCLES)
                peak_valu = N_SIMDS * clk * 1e9 / (vpu * ISSUE_CYCLES)
                ups = per_launch_updates / avg_kernel_s
                issue = {"instructions_per_update": ipu, "valu_per_update": vpu, "salu_per_update": ij.get("salu_per_update"),
                         "cycles_per_instruction": ISSUE_CYCLES, "simds": N_SIMDS, "clock_ghz": clk,
                         "peak_updates_per_s": peak, "achieved_updates_per_s": ups, "frac": ups / peak,
                         "valu_only_peak_updates_per_s": peak_valu, "valu_only_frac": ups / peak_valu,
                         "source": ij.get("_file", "profiles/r04_issue.json"), "measured_on_build": ij.get("build"),
                         "current_build": cur}
            steady, cur = tagged("r04_steady_state.json")
            if steady:
                steady = dict(steady, current_build=cur)
        if steady_live is not None:
            steady = dict(steady or {}, live=steady_live)
        roofline = {
            "bound": "hbm", "kernel": "sweep_fast_kernel" if args.rng == "philox" else "sweep_kernel<mt19937-compat>", "achieved": achieved, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
            "alg_bytes_per_update": balg, "updates_per_launch": per_launch_updates,
            "avg_launch_ms": avg_kernel_s * 1e3,
            "note": "bound/achieved/frac are the SURVEY 8(d) HBM figure (algorithmic bytes); by its counters the "
                    "kernel is instruction-issue/latency-bound, not HBM-bound: see issue_bound",
            "issue_bound": issue,
        }
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
            "data": "synthetic" if not args.edgelist else "edge list file (the reference's shipped data set)" if args.preset else "edge list file",
            "sweeps_per_s": total_updates / elapsed / n,
            "config": {
                "workload": ("BASELINE configs[2]: " if default_cfg else (args.preset + ": ") if args.preset else "custom: ")
                + ("edge list %s, " % os.path.basename(args.edgelist) if args.edgelist else "planted bipartite ")
                + "N_a=%d N_b=%d E=%d Ka=%d Kb=%d, %d chains/GPU, constant T=1, eps=1, "
                  "%s start, %s mode%s" % (na, nb, E, ka, kb, args.chains, "planted-partition" if args.planted_start else "randomised",
                                                       "Philox" if args.rng == "philox" else "mt19937-compat (bit-exact with the reference)",
                                                       (", node ids renumbered at random" +
                                                        (", locality reordering at ingest (%.1f s)" % reorder_s if reorder else ""))
                                                       if args.shuffle_ids else ""),
                "chains_total": args.chains * world, "step": ("one sweep (n node updates) of every chain" if args.sweeps_per_step == 1
                                                          else "%d sweeps (of n node updates) of every chain in one launch" % args.sweeps_per_step),
                "spinup_sweeps_before_warmup": len(spin_ms),
                "accepted_fraction_last_timed_sweep": accepted_frac,
                "steps_per_pass_of_the_timed_launches": {str(k): pass_steps.count(k) for k in sorted(set(pass_steps))},
                # every timed step in order: steps per pass the library chose for it (bisbm_pass_policy.hpp: the incumbent depth,
                # one look at a neighbour in sixteen launches), its kernel time, its accepted fraction
                "per_launch_steps_per_pass": pass_steps, "per_launch_ms": launch_ms, "per_launch_accepted": launch_acc,
                "parallelism": "chains sharded, no collective in the sweep path",
                "collective_backend": (backend if world > 1 else None), "collective_backend_note": backend_note,
            },
            "build": tag,
            "roofline": roofline,
            "equilibrated_start": equil,
            "steady_state": steady,
        }
        out.update(extras)
        if world == 1 and not args.no_cpu_baseline:
            del model
            out["cpu_baseline"] = cpu_baseline(n, E, probe=REF_PROBE_N1000 if args.preset == "BASELINE configs[1]" else None)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
