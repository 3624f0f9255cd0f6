#!/bin/bash
# round 3: the profile set of the final build (run on the GPU box from the repo root; results under gpurun_out/prof_$1/)
#   tools/r03_final_profiles.sh TAG
TAG=${1:-v46}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "default line done"
tools/profile.sh $TAG > $OUT/profile_sh.log 2>&1; echo "rocprofv3 passes done"
python bench.py --spinup 150 --no-cpu-baseline --no-extras > $OUT/bench_after_150_sweeps.json 2> $OUT/bench_after_150.err; echo "after-150 done"
python bench.py --shuffle-ids --no-reorder --no-cpu-baseline --no-extras > $OUT/bench_shuffled_ids.json 2>/dev/null; echo "shuffled done"
python bench.py --shuffle-ids --no-cpu-baseline --no-extras > $OUT/bench_shuffled_ids_reordered.json 2>/dev/null; echo "reordered done"
python bench.py --na 2000000 --nb 2000000 --edges 50000000 --ka 64 --kb 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_config5_shape_1gpu.json 2>/dev/null; echo "config5 shape done"
python bench.py --edgelist n_1000 > $OUT/bench_config2_n_1000.json 2>/dev/null; echo "n_1000 done"
python tools/debug/schedule_speed.py > $OUT/schedule_speed.txt 2>&1; python tools/debug/schedule_speed.py await >> $OUT/schedule_speed.txt 2>&1; echo "schedules done"
python tools/debug/small_graph_speed.py > $OUT/small_graph_speed.txt 2>&1; echo "small graphs done"
ls $OUT
