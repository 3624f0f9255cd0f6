#!/bin/bash
# rocprofv3 evidence for bench.py: kernel trace + stats, then one PMC pass per counter group (separate runs, no
# tracing combined with --pmc).  Run from the repo root on the GPU box; results under gpurun_out/prof_<tag>/.
#   tools/profile.sh TAG [bench args...]
set -e
TAG=${1:-run}; shift || true
ROOT=$(pwd); OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-extras $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -o kt -- python3 bench.py --steps 6 --warmup 2 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/kt.err
find $OUT/kt -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# the counter passes run the kind of pass the bench line's timed launches run (two steps per pass at 32 + 32 blocks from a
# randomised start: config.steps_per_pass_of_the_timed_launches); left free, the first launches of a process try every depth
export BISBM_PASS_DEPTH=${PMC_PASS_DEPTH:-2}
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_INSTS_SMEM" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -o pmc -- python3 bench.py --steps 1 --warmup 0 $ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err || echo "pass $i ($grp) failed"
  echo "pass $i done: $grp"
done
python3 tools/summarize_pmc.py $OUT/pmc_counters.csv sweep_fast $OUT/pmc[0-9]* > /dev/null
rm -rf $OUT/kt/*/*.db 2>/dev/null || true
cat $OUT/pmc_counters.csv
