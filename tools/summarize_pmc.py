#!/usr/bin/env python3
"""Sum rocprofv3 counter_collection CSVs per (pass, kernel, counter) for the kernels whose name matches.
usage: summarize_pmc.py OUT.csv KERNEL_SUBSTRING PASSDIR [PASSDIR ...]"""
import csv, glob, os, sys
from collections import defaultdict

out, needle, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
rows = []
for d in dirs:
    acc, launches = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if needle not in r["Kernel_Name"]:
                    continue
                key = (r["Kernel_Name"], r["Counter_Name"])
                acc[key] += float(r["Counter_Value"])
                launches[key].add(r.get("Dispatch_Id", r.get("Correlation_Id", "")))
    for (k, c), v in sorted(acc.items()):
        rows.append((os.path.basename(d.rstrip("/")), k, c, v, len(launches[(k, c)])))
with open(out, "w", newline="") as fh:
    w = csv.writer(fh, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["pass", "kernel", "counter", "value_sum_over_launches", "launches"])
    w.writerows(rows)
print(open(out).read())
