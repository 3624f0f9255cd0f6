#!/bin/bash
# One rocprofv3 --pmc pass per counter group given on the command line (quoted, space separated), for
# `bench.py $BENCH_ARGS`; sums for the sweep kernel land in gpurun_out/pmc_<TAG>/pmc_counters.csv.
#   BENCH_ARGS="--chains 256" tools/pmc.sh TAG "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "TA_BUSY_avr ..."
TAG=$1; shift
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT; export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc$i -o pmc -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras $BENCH_ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err || echo "pass $i ($grp) failed"
done
python3 tools/summarize_pmc.py $OUT/pmc_counters.csv sweep_fast $OUT/pmc[0-9]* > /dev/null
python3 -c "import csv,sys; [print(r[0], r[2], r[3]) for r in csv.reader(open(sys.argv[1]))]" $OUT/pmc_counters.csv
