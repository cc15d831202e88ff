#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_cooc_$1; shift
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu --no-itemsim --no-regime --no-factorization $*"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY -d $OUT/a -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/a.err || exit 1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/b -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/b.err || exit 2
for p in a b; do python3 tools/prof_summary.py $OUT/$p k_cooc; done
find $OUT -name "*.csv" -delete
