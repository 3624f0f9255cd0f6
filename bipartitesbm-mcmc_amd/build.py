"""Builds libbisbm_hip.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

-ffp-contract=off is part of the contract, not a tuning flag: the FP64 expressions of the dS /
Hastings path must round exactly like the host code they are compared with (no FMA contraction)."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libbisbm_hip.so")
# translation units of the library (csrc/): the device code (kernels + their launchers), the host side of the C ABI
# (bisbm_engine.hpp describes which unit holds what), plain C++ for the tables and the text / CSR ingest
SOURCES = ["bisbm_kernels.hip", "bisbm_sweep_fast.hip", "bisbm_handle.hip", "bisbm_anneal.hip", "bisbm_marginals.hip",
           "bisbm_multi.hip", "bisbm_merge.hip", "bisbm_tables.cpp", "bisbm_io.cpp"]
HEADERS = ["bisbm_device.hpp", "bisbm_kernels.hpp", "bisbm_engine.hpp", "bisbm_pass_policy.hpp", os.path.join("..", "host", "bisbm.hpp"),
           os.path.join("..", "host", "mcmc_main.cpp"), os.path.join("..", "..", "include", "bisbm.h"),
           os.path.join("..", "..", "include", "bisbm_io.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread",
         "-Wall", "-Wno-unused-function",
         # The step loop is a tree of wave-uniform branches (scalar compares, ballots).  Left to its default the
         # backend structurizes the whole loop -- flag registers, chains of always-taken jumps and a copy of every
         # loop-carried register on every path (~40 issue slots per step); told to leave uniform regions alone it
         # emits the branches as written and updates the block state in place.
         "-mllvm", "-structurizecfg-skip-uniform-regions=true",
         # single-lane LDS atomics (the early-stop bookkeeping below T = 1) stay single instructions
         "-mllvm", "-amdgpu-atomic-optimizer-strategy=None",
         # loop heads on 64-byte boundaries (the instruction cache's line): the pass loop of the production kernel is entered by a
         # taken branch ~2 million times per chain and sweep, and where its head falls in a line is worth 1 % (same-box A/B with 32 /
         # 64 / 128: profiles/r04_ab_pass_scheduling.txt) -- and it takes the placement noise out of every later A/B
         "-falign-loops=64"]


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the engine is HIP-only and cannot be built without ROCm")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile the library (and the CLI) in-tree.  Safe to call from several processes at once (the ranks of a
    multi-GPU job on a fresh checkout): one holds the lock and compiles, the others wait and find the result."""
    import fcntl
    if not force and not needs_build():
        return LIB
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


RESOURCES = os.path.join(HERE, "libbisbm_hip.resources.json")  # registers / spills / scratch per kernel of the built library


def _resource_usage(remarks):
    """{kernel: {"VGPRs": n, "SGPRs Spill": n, "VGPRs Spill": n, "ScratchSize [bytes/lane]": n, ...}} from the
    compiler's -Rpass-analysis=kernel-resource-usage remarks (they do not change the generated code)."""
    import re
    out, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+) \[-Rpass-analysis", line)
        if m and cur is not None:
            v = m.group(2)
            cur[m.group(1).strip()] = int(v) if v.isdigit() else v
    return out


def compile_library(out, extra=(), verbose=False, jobs=None):
    """Every translation unit to an object file (side by side: the production kernel's unit takes most of the time), then one
    link.  Returns the compiler's remarks (stderr of all units).  `extra`: more hipcc flags (diagnostic builds, tools/)."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    cc = hipcc()
    objdir = tempfile.mkdtemp(prefix="bisbm_obj_")
    try:
        # the production kernel's 20 variants: four units of five kernels + the dispatch (BISBM_FAST_PART, bisbm_sweep_fast.hip);
        # a build with in-kernel stamps keeps them in one unit (its counters are one device symbol)
        units = []
        for src in SOURCES:
            if src == "bisbm_sweep_fast.hip" and not any("BISBM_STAMPS" in f for f in extra):
                units += [(src, ["-DBISBM_FAST_PART=%d" % part], "%s.part%d.o" % (src, part)) for part in range(5)]
            else:
                units.append((src, [], src + ".o"))

        def one(unit):
            src, defs, objname = unit
            obj = os.path.join(objdir, objname)
            cmd = [cc] + FLAGS + ["-Rpass-analysis=kernel-resource-usage"] + list(extra) + defs + ["-c", "-o", obj, os.path.join(CSRC, src)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s%s" % (src, r.stdout, r.stderr))
            return obj, r.stderr
        with ThreadPoolExecutor(max_workers=jobs or min(len(units), os.cpu_count() or 1)) as pool:
            done = list(pool.map(one, units))
        tmp = out + ".tmp%d" % os.getpid()
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", tmp] + [o for o, _ in done]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("linking %s failed:\n%s%s" % (out, r.stdout, r.stderr))
        os.replace(tmp, out)  # (a process that has the old library mapped keeps its inode)
        return "".join(err for _, err in done)
    finally:
        shutil.rmtree(objdir, ignore_errors=True)


def _build_locked(verbose):
    import json
    extra = os.environ.get("BISBM_EXTRA_HIPCC_FLAGS", "").split()  # diagnostic builds (e.g. -DBISBM_ABLATE=1)
    remarks = compile_library(LIB, extra, verbose)
    usage = _resource_usage(remarks)
    with open(RESOURCES, "w") as fh:
        json.dump(usage, fh, indent=1, sort_keys=True)
    if verbose:
        rest = "\n".join(l for l in remarks.splitlines() if "kernel-resource-usage" not in l and not l.lstrip().startswith(("|", "1")))
        if rest.strip():
            print(rest, file=sys.stderr)
        print("%d kernels; vector registers spilled: %d" % (len(usage), sum(k.get("VGPRs Spill", 0) for k in usage.values())), file=sys.stderr)
    build_cli(verbose=verbose)
    return LIB


CLI = os.path.join(HERE, "bin", "mcmc")


def build_cli(verbose=False):
    """The reference's command line (host/mcmc_main.cpp) linked against the in-tree library."""
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    cmd = [hipcc(), "-O2", "-std=c++17", "-o", CLI, os.path.join(HERE, "host", "mcmc_main.cpp"),
           "-L" + HERE, "-lbisbm_hip", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building bin/mcmc failed:\n" + r.stdout + r.stderr)
    return CLI


if __name__ == "__main__":
    # python build.py [--force]                      the product library (+ bin/mcmc)
    # python build.py --variant OUT.so [flags ...]   a diagnostic build beside it (tools/build_variant.sh)
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        compile_library(os.path.abspath(sys.argv[i + 1]), sys.argv[i + 2:], verbose=False)
        print(sys.argv[i + 1])
    else:
        print(build(force="--force" in sys.argv, verbose=True))
