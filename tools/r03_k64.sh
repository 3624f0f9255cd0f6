#!/bin/bash
# round 3: config-5 shape (Ka = Kb = 64) bench line -> gpurun_out/$1.json
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python bench.py --na 2000000 --nb 2000000 --edges 50000000 --ka 64 --kb 64 --steps ${STEPS:-2} --warmup 1 --no-cpu-baseline --no-extras > gpurun_out/$1.json 2> gpurun_out/$1.err
python - gpurun_out/$1.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("K64 shape: %.4e updates/s, %.1f ms per sweep" % (d["value"], d["ms_per_step"]))
PY
