"""AddressSanitizer + UndefinedBehaviorSanitizer runs of everything on the path that executes on the CPU (the GPU pool has no
sanitizer support; the kernels are covered by the parity tests instead):
  * the product's host I/O (bisbm_io.cpp: text scanners, CSR cache, renumbering) through tests/native/sanitize_io.cpp,
  * the re-hosted CLI's argument handling (host/mcmc_main.cpp + host/bisbm.hpp), up to the point where it asks for a GPU,
  * the CPU checker itself (oracle/bisbm_oracle.c), by running its own test module against a sanitizer build of it.
CPU only; g++/gcc from the image."""
import os
import random
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "bipartitesbm-mcmc_amd")
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:exitcode=97", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def _gcc_file(name):
    return subprocess.run(["gcc", "-print-file-name=" + name], check=True, capture_output=True, text=True).stdout.strip()


def _clean(proc):
    err = proc.stderr if isinstance(proc.stderr, str) else proc.stderr.decode(errors="replace")
    assert "AddressSanitizer" not in err and "runtime error:" not in err and "LeakSanitizer" not in err, err[-3000:]
    assert proc.returncode != 97 and proc.returncode >= 0, (proc.returncode, err[-2000:])  # 97 = sanitizer, < 0 = signal


@pytest.fixture(scope="module")
def san_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("san"))


def test_host_io_under_sanitizers(san_dir):
    exe = os.path.join(san_dir, "sanitize_io")
    subprocess.run(["g++", "-std=c++17", *SAN, "-pthread", "-o", exe, os.path.join(ROOT, "tests", "native", "sanitize_io.cpp"),
                    os.path.join(PKG, "csrc", "bisbm_io.cpp")], check=True, capture_output=True)
    for seed in (1, 2, 3):
        work = os.path.join(san_dir, "io%d" % seed)
        os.makedirs(work, exist_ok=True)
        p = subprocess.run([exe, work, str(seed)], capture_output=True, text=True, env=ENV, timeout=300)
        _clean(p)
        assert p.returncode == 0, p.stderr[-2000:]


def test_cli_argument_handling_under_sanitizers(san_dir):
    """Every invocation ends before or at bisbm_create (no GPU here: exit 3) -- what is checked is that no argv, however
    malformed, makes the host shell read or write out of bounds on the way there."""
    import importlib
    importlib.import_module("bipartitesbm-mcmc_amd.build").build()  # libbisbm_hip.so to link against
    exe = os.path.join(san_dir, "mcmc_san")
    subprocess.run(["g++", "-std=c++17", *SAN, "-o", exe, os.path.join(PKG, "host", "mcmc_main.cpp"), "-L" + PKG, "-lbisbm_hip",
                    "-Wl,-rpath," + PKG], check=True, capture_output=True)
    el = os.path.join(ROOT, "tests", "golden", "southernWomen.edgelist")
    good = ["-e", el, "-y", "18", "14", "-n", "4", "4", "4", "3", "3", "3", "3", "3", "3", "2", "-z", "5", "5", "-t", "3200", "-x", "100",
            "-c", "exponential", "-a", "10", "0.1", "-E", "0.001", "--randomize", "-d", "1"]
    cases = [good, [], ["-h"], ["--help"], ["-e"], ["-e", "/nonexistent"], ["-y"], ["-y", "18"], ["-n"], ["-z", "5"],
             good + ["--chains"], good + ["--chains", "0"], good + ["--chains", "-4"], good + ["--chains", "99999999999999999999"],
             good + ["--rng", "nonsense"], good + ["--rng"], good + ["-c", "nonsense"], good + ["-a"], good + ["-t", "abc"],
             good + ["-t", "-1"], good + ["-E", "nan"], good + ["-d", "1e99"], good + ["--merge"], good + ["--merge", "--nature"],
             good + ["--membership_path", "/nonexistent"], good + ["--csr_cache", "--reorder"], good + ["--marginalize"], good + ["--marginalize", "-b", "x", "-f", "0"], good + ["--unknown-flag", "3"],
             ["-y", "1", "1", "-n", "1", "1", "-z", "1", "1", "-e", el],  # ids beyond n
             ["-e", el, "-y", "18", "14", "-n", "30", "2", "-z", "1", "1", "-t", "10"],  # block sizes that do not add up
             ["-e", el, "-y", "0", "0", "-n", "-z", "0", "0"], ["-e", el, "-y", "18", "14", "-z", "300", "5", "-n"] + ["1"] * 305]
    rng = random.Random(7)
    vocab = ["-e", el, "-y", "-n", "-z", "-t", "-x", "-c", "-a", "-E", "-d", "--randomize", "--merge", "--nature", "--chains", "--rng",
             "philox", "mt19937-compat", "--maximize", "-b", "-f", "--membership_path", "--csr_cache", "--reorder", "--marginalize", "0", "1", "5", "18", "14",
             "-1", "1e9", "", "constant", "linear", "abrupt_cool", "logarithmic", "exponential", "x" * 300]
    for _ in range(60):
        cases.append([rng.choice(vocab) for _ in range(rng.randint(1, 25))])
    for argv in cases:
        p = subprocess.run([exe] + argv, capture_output=True, env=ENV, timeout=120)
        _clean(p)
        assert p.returncode in (0, 1, 2, 3), (argv, p.returncode, p.stderr[-500:])


def test_cpu_checker_under_sanitizers(san_dir):
    so = os.path.join(san_dir, "liboracle_san.so")
    subprocess.run(["gcc", "-std=gnu11", "-fPIC", "-ffp-contract=off", *SAN, "-shared", "-o", so,
                    os.path.join(ROOT, "oracle", "bisbm_oracle.c"), "-lm"], check=True, capture_output=True)
    env = dict(ENV, BISBM_ORACLE_SO=so, LD_PRELOAD=_gcc_file("libasan.so") + ":" + _gcc_file("libubsan.so"),
               ASAN_OPTIONS="detect_leaks=0:exitcode=97")  # the interpreter's own allocations are not the subject
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_oracle.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=env, cwd=ROOT, timeout=900)
    _clean(p)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
