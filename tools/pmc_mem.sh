#!/bin/bash
# memory-path counters of the scoring kernel (GPU box)
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_mem_$1; shift
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu --no-itemsim $*"
rocprofv3 --pmc TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum -d $OUT/a -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/a.err || exit 1
rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum -d $OUT/b -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/b.err || exit 2
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/c -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/c.err || exit 3
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum TCC_READ_sum -d $OUT/d -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/d.err || tail -3 $OUT/d.err
for p in a b c d; do python3 tools/prof_summary.py $OUT/$p k_score; done
find $OUT -name "*.csv" -delete
