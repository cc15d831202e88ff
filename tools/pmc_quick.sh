#!/bin/bash
# quick PMC look at one configuration (GPU box): usage tools/pmc_quick.sh <tag> [bench args]; env is inherited
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu --no-itemsim $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 bench.py $ARGS > $OUT/bench.json 2> $OUT/trace.err || exit 1
rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/fetch.err || exit 2
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/tcc -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/tcc.err || exit 4
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM -d $OUT/sq -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/sq.err || exit 5
for p in trace fetch tcc sq; do python3 tools/prof_summary.py $OUT/$p fy:: > $OUT/summary_$p.txt; done
find $OUT -name "*.csv" ! -name "*stats*" -delete
cat $OUT/summary_*.txt | grep -E "k_score|k_topn|k_cooc|total kernel"
