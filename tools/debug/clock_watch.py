"""Samples the GPU's clock / power sysfs nodes every 50 ms while a command runs (diagnostic for the two speeds of one and
the same sweep launch).  Reads sysfs only -- this process never touches the GPU.

    python tools/debug/clock_watch.py OUT.txt -- python bench.py --no-cpu-baseline
"""
import glob
import os
import subprocess
import sys
import time


def nodes():
    out = []
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        for name in ("pp_dpm_sclk", "pp_dpm_mclk", "pp_dpm_fclk", "pp_dpm_socclk", "gpu_busy_percent", "mem_busy_percent"):
            p = os.path.join(dev, name)
            if os.path.exists(p):
                out.append(p)
        for hw in glob.glob(os.path.join(dev, "hwmon", "hwmon*")):
            for name in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input"):
                p = os.path.join(hw, name)
                if os.path.exists(p):
                    out.append(p)
    return out


def read(p):
    try:
        with open(p) as f:
            t = f.read().strip()
    except OSError as e:
        return "ERR(%s)" % e.errno
    if "\n" in t or "*" in t:  # pp_dpm_*: keep the starred (current) level
        cur = [ln for ln in t.splitlines() if ln.endswith("*")]
        return cur[0].split(":")[1].strip(" *") if cur else t.replace("\n", "|")
    return t


def main():
    out_path, cmd = sys.argv[1], sys.argv[sys.argv.index("--") + 1:]
    ns = nodes()
    with open(out_path, "w") as out:
        out.write("# nodes:\n" + "".join("#  %d %s\n" % (i, p) for i, p in enumerate(ns)))
        child = subprocess.Popen(cmd)
        t0 = time.time()
        last = None
        while child.poll() is None:
            vals = [read(p) for p in ns]
            if vals != last:  # only changes
                out.write("%8.3f %s\n" % (time.time() - t0, " ".join(vals)))
                out.flush()
                last = vals
            time.sleep(0.05)
        out.write("# exit %d after %.1f s (t0 = %.3f)\n" % (child.returncode, time.time() - t0, t0))
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
