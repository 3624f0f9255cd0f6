#!/bin/bash
# Known-bytes calibration of FETCH_SIZE / WRITE_SIZE (tools/probe/fetch_calib.hip): one rocprofv3 --pmc pass each,
# per-kernel counters against the byte counts the program states -> gpurun_out/fetch_calib/summary.json
set -e
cd "$(dirname "$0")/.."
OUT=$(pwd)/gpurun_out/fetch_calib; mkdir -p $OUT; export TMPDIR=/tmp
[ -x tools/probe/bin/fetch_calib ] || hipcc --offload-arch=gfx950 -O3 -o tools/probe/bin/fetch_calib tools/probe/fetch_calib.hip
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- tools/probe/bin/fetch_calib all > $OUT/stated_bytes.txt 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o pmc -- tools/probe/bin/fetch_calib store16 > $OUT/stated_bytes_write.txt 2> $OUT/write.err
python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
GiB2 = 2 << 30
stated = {"stream16": GiB2, "stream4": GiB2, "rows64": GiB2, "store16": GiB2}
res = {}
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            res.setdefault(k, {}).setdefault(counter + "_KB", 0.0)
            res[k][counter + "_KB"] += float(r["Counter_Value"])
            res[k].setdefault("dispatches", 0)
            res[k]["dispatches"] += 1
summary = {"_what": "rocprofv3 counters (unit KB, as reported) against bytes known by construction; buffer 2 GiB, every "
                    "byte / line touched once (tools/probe/fetch_calib.hip)", "kernels": res, "stated_bytes": stated,
           "gather1": {"loads": GiB2 // 128, "note": "dispatch 1 of gather1: one byte per 128-B line; dispatch 2: one byte per 64-B sector"}}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
