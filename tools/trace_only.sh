#!/bin/bash
# kernel trace of one bench configuration: tools/trace_only.sh <tag> [bench args...]  -> gpurun_out/trace_<tag>/summary_trace.txt
set -o pipefail
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/trace_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t --output-format csv -- python3 bench.py --steps 1 --warmup 1 --no-cpu --no-regime --no-itemsim --no-factorization "$@" > $OUT/bench.json 2> $OUT/trace.err || exit 1
python3 tools/prof_summary.py $OUT/trace > $OUT/summary_trace.txt
find $OUT -name "*.csv" ! -name "*stats*" -delete
head -40 $OUT/summary_trace.txt
