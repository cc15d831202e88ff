#!/bin/bash
# Runs on the GPU box (via gpurun): kernel trace + the PMC passes of MI355X_MICROARCH.md (separate runs), summaries into gpurun_out/.
# usage: tools/profile_round.sh <tag> [bench args...]
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
ARGS="--steps 1 --warmup 1 --no-cpu --no-regime $*"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err || exit 1
python3 tools/prof_summary.py $OUT/trace > $OUT/summary_trace.txt
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_fetch.err || exit 2
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_write.err || exit 3
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcc -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_tcc.err || exit 4
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_sq.err || exit 5
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc_lds -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_lds.err || exit 6
# CU-side traffic (L1 -> L2 read requests): what the scoring kernels, whose panels never leave the chip, actually move
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum -d $OUT/pmc_l2 -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/pmc_l2.err || exit 7
for p in pmc_fetch pmc_write pmc_tcc pmc_sq pmc_lds pmc_l2; do PROF_TOP=60 python3 tools/prof_summary.py $OUT/$p > $OUT/summary_$p.txt; done
PROF_TOP=60 python3 tools/prof_summary.py $OUT/trace > $OUT/summary_trace.txt
# keep only the summaries + the stats csv (the raw traces are large)
find $OUT -name "*.csv" ! -name "*stats*" -delete
cat $OUT/summary_trace.txt; grep -h k_score $OUT/summary_pmc_*.txt
