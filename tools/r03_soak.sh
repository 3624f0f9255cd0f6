#!/bin/bash
# round 3: long-run consistency of the final build (results: gpurun_out/r03_soak/)
cd "$(dirname "$0")/.."
OUT=gpurun_out/r03_soak; mkdir -p $OUT
python -m pytest tests -m gpu -x -q > $OUT/gpu_tests.log 2>&1; tail -3 $OUT/gpu_tests.log
(python tools/soak.py bench 100; python tools/soak.py bench 48 linear 1.2 1e-8; python tools/soak.py bench 48 abrupt_cool 2.0e7;
 python tools/soak.py config5 24; python tools/soak.py config5 12 exponential 2.0 0.99999995; python tools/soak.py n_1000 6000) > $OUT/soak.txt 2>&1
grep -v "^  " $OUT/soak.txt
BISBM_FUZZ_SEEDS=${FUZZ:-1200} python -m pytest tests/test_gpu_fuzz.py -x -q > $OUT/fuzz.log 2>&1; tail -3 $OUT/fuzz.log
