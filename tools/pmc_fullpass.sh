#!/bin/bash
# Counter passes over a job that takes the PLAIN FULL PASS (k_score<4, true, 8> + k_topn_long), e.g. the reference's default configuration:
#   tools/pmc_fullpass.sh k50_n1000 --clusters 50 --top-n 1000          -> gpurun_out/pmc_fullpass_<tag>/summary_*.txt
# Separate rocprofv3 --pmc runs (never combined with a trace); a pass that fails leaves its .err and the others still run.
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_fullpass_$TAG
mkdir -p $OUT
ARGS="--steps 1 --warmup 0 --no-cpu --no-regime --no-itemsim --no-factorization $*"
PER=${PMC_PASS_TIMEOUT:-55}
ONLY=${PMC_PASSES:-}      # e.g. PMC_PASSES="fetch write": only these passes
pass() {   # name, counters...
    local name=$1; shift
    if [ -n "$ONLY" ] && ! echo " $ONLY " | grep -q " $name "; then return; fi
    timeout -k 10 $PER rocprofv3 --pmc "$@" -d $OUT/$name -o p --output-format csv -- python3 bench.py $ARGS > /dev/null 2> $OUT/$name.err
    local rc=$?
    PROF_TOP=60 python3 tools/prof_summary.py $OUT/$name fy:: > $OUT/summary_$name.txt 2>> $OUT/$name.err
    find $OUT/$name -name "*.csv" -delete
    echo "pass $name rc $rc"
    # a pass killed at its limit has told us something: no further GPU step in this call
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
}
pass sq_issue SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass l2 TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass sq_mem SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS
pass fetch FETCH_SIZE
pass write WRITE_SIZE
# (a pass with TA_* / TCP_PENDING_STALL counters aborted inside rocprofv3 on this configuration -- signal 6, round 4 -- and is not run)
grep -h "k_score<\|k_topn_long\|k_topn_select\|k_cooc_rm2<" $OUT/summary_*.txt
