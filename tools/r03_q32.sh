#!/bin/bash
# round 3: two against four steps per pass at 32 + 32 blocks (pinned depth), three regimes of the bench workload
OUT=gpurun_out/${1:-q32}; mkdir -p $OUT
for d in 2 4; do
  BISBM_PASS_DEPTH=$d python bench.py --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_d$d.json 2>$OUT/bench_d$d.err
  BISBM_PASS_DEPTH=$d python bench.py --steps 6 --warmup 2 --spinup 150 --no-cpu-baseline --no-extras > $OUT/bench150_d$d.json 2>$OUT/bench150_d$d.err
  python - <<PY
import json
a=json.loads(open("$OUT/bench_d$d.json").read().strip().splitlines()[-1]); b=json.loads(open("$OUT/bench150_d$d.json").read().strip().splitlines()[-1])
print("depth", $d, "random start %.4g" % a["value"], "after 150 sweeps %.4g" % b["value"], "equilibrated %.4g" % a["equilibrated_start"]["updates_per_s_per_gpu"])
PY
done
