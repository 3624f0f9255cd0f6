#!/bin/bash
# round 3, first GPU call: cross-wave hand-off cost, stage stamps of the K <= 32 pass and of the K = 64 step
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_first; mkdir -p $OUT
LIB=bipartitesbm-mcmc_amd/libbisbm_hip.so
tools/probe/bin/wave_handoff > $OUT/wave_handoff.txt 2>&1
cp $LIB /tmp/lib_keep.so
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_base.json 2> $OUT/bench_base.err
python bench.py --na 2000000 --nb 2000000 --edges 50000000 --ka 64 --kb 64 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/bench_k64.json 2> $OUT/bench_k64.err
cp ab/stamps1.so $LIB
(echo "== K32 pair"; python bench.py --chains 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamps | tail -12
 echo "== K32 single"; BISBM_SINGLE_STEPS=1 python bench.py --chains 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamps | tail -12
 echo "== K64"; python bench.py --na 2000000 --nb 2000000 --edges 50000000 --ka 64 --kb 64 --chains 1024 --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2>&1 >/dev/null | grep stamps | tail -12) > $OUT/stamps.txt 2>&1
cp /tmp/lib_keep.so $LIB
cat $OUT/wave_handoff.txt $OUT/stamps.txt; tail -c 600 $OUT/bench_base.json; tail -c 600 $OUT/bench_k64.json
