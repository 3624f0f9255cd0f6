#!/bin/bash
# round 4: the profile set of a build (run on the GPU box from the repo root; results under gpurun_out/prof_$TAG/).
#   tools/r04_profiles.sh TAG STAGE     STAGE 1: bench lines; 2: rocprofv3 sets (default workload, config-5 shape)
TAG=${1:-c1}; STAGE=${2:-1}
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_$TAG; mkdir -p $OUT
if [ "$STAGE" = "1" ]; then
  python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "default line done"
  BISBM_PASS_LOG=1 BISBM_BENCH_VERBOSE=1 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver_protocol.json 2> $OUT/bench_driver_protocol.err; echo "driver protocol done"
  python3 bench.py --na 2000000 --nb 2000000 --edges 50000000 --ka 64 --kb 64 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_config5_shape_1gpu.json 2>/dev/null; echo "config5 shape done"
  python3 bench.py --edgelist n_1000 > $OUT/bench_config2_n_1000.json 2>/dev/null; echo "n_1000 done"
  python3 bench.py --edgelist n_1000 --rng compat --chains 1024 --no-cpu-baseline --no-extras > $OUT/bench_compat_n_1000_1024_chains.json 2>/dev/null; echo "compat done"
  python3 bench.py --shuffle-ids --no-reorder --no-cpu-baseline --no-extras > $OUT/bench_shuffled_ids.json 2>/dev/null; echo "shuffled done"
  python3 bench.py --shuffle-ids --no-cpu-baseline --no-extras > $OUT/bench_shuffled_ids_reordered.json 2>/dev/null; echo "reordered done"
else
  tools/profile.sh $TAG > $OUT/profile_sh.log 2>&1; echo "rocprofv3 passes (default workload) done"
  tools/profile.sh ${TAG}_config5 --na 2000000 --nb 2000000 --edges 50000000 --ka 64 --kb 64 --spinup 1 > $OUT/profile_config5_sh.log 2>&1; echo "rocprofv3 passes (config-5 shape) done"
fi
ls $OUT
