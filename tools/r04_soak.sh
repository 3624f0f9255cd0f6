#!/bin/bash
# round 4: long-run consistency of the kept build (results: gpurun_out/r04_soak/).  The production path (launches that cannot
# stop early take the running sum from the description length: the state checks are what counts there) and, with
# BISBM_KEEP_SUM=1, the step-by-step sum against the change of the description length.
cd "$(dirname "$0")/.."
OUT=gpurun_out/r04_soak; mkdir -p $OUT
(echo "== BISBM_KEEP_SUM=0 (production path)"; BISBM_KEEP_SUM=0 python3 tools/soak.py bench 80; BISBM_KEEP_SUM=0 python3 tools/soak.py bench 40 linear 1.2 1e-8;
 BISBM_KEEP_SUM=0 python3 tools/soak.py config5 16; BISBM_KEEP_SUM=0 python3 tools/soak.py n_1000 6000;
 echo "== BISBM_KEEP_SUM=1 (step-by-step sum)"; python3 tools/soak.py bench 60; python3 tools/soak.py bench 40 abrupt_cool 2.0e7; python3 tools/soak.py config5 12 exponential 2.0 0.99999995) > $OUT/soak.txt 2>&1
grep -v "^  " $OUT/soak.txt
BISBM_FUZZ_SEEDS=${FUZZ:-2000} python3 -m pytest tests/test_gpu_fuzz.py -x -q > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
BISBM_LAUNCH_STEPS=1 BISBM_FUZZ_SEEDS=600 python3 -m pytest tests/test_gpu_fuzz.py -x -q > $OUT/fuzz_single_sweep_launches.log 2>&1; tail -3 $OUT/fuzz_single_sweep_launches.log
