#!/bin/bash
# First GPU call of the next round, in one piece (about 13 minutes; give gpurun --timeout 1100):
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/round5_first_call.sh > gpurun_out/r5_first.log 2>&1; tail -40 gpurun_out/r5_first.log'
# 1. the two experiments that were written without GPU time (tools/micro/score_loop.hip incl. its lambda = 0 mode, tools/micro/score_tiled.hip
#    with what each arrangement fetches); 2. the prepared patches applied ON THE BOX'S COPY of the tree (nothing comes back but gpurun_out/),
#    the library rebuilt, the gating tests, the bench line.  A step that fails or is killed at its limit ends the call: no GPU step after it.
set -o pipefail
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r5_first
mkdir -p $OUT
fail() { echo "FAILED: $*"; exit 1; }
step() { echo; echo "== $*"; }

step "micro: the row loop (tools/micro/score_loop.hip)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/score_loop tools/micro/score_loop.hip 2> $OUT/score_loop.build || fail "score_loop.hip does not compile"
timeout -k 10 60 /tmp/score_loop 4096 4096 20 > $OUT/score_loop.txt 2>&1; rc=$?; cat $OUT/score_loop.txt; [ $rc = 0 ] || fail "score_loop (rc $rc)"
timeout -k 10 60 /tmp/score_loop 4096 4096 20 zeros > $OUT/score_loop_zeros.txt 2>&1; rc=$?; cat $OUT/score_loop_zeros.txt; [ $rc = 0 ] || fail "score_loop zeros (rc $rc)"
timeout -k 10 90 /tmp/score_loop 16384 3250 20 > $OUT/score_loop_16k.txt 2>&1; rc=$?; cat $OUT/score_loop_16k.txt; [ $rc = 0 ] || fail "score_loop 16k (rc $rc)"

step "micro: one XCD per chunk, row blocks outermost (tools/micro/score_tiled.hip)"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/score_tiled tools/micro/score_tiled.hip 2> $OUT/score_tiled.build || fail "score_tiled.hip does not compile"
timeout -k 10 150 /tmp/score_tiled > $OUT/score_tiled.txt 2>&1; rc=$?; cat $OUT/score_tiled.txt; [ $rc = 0 ] || fail "score_tiled (rc $rc)"
timeout -k 10 150 /tmp/score_tiled 32768 3250 2048 > $OUT/score_tiled_rb2048.txt 2>&1; rc=$?; cat $OUT/score_tiled_rb2048.txt; [ $rc = 0 ] || fail "score_tiled RB 2048 (rc $rc)"
timeout -k 10 150 rocprofv3 --pmc FETCH_SIZE -d $OUT/tiled_fetch -o p --output-format csv -- /tmp/score_tiled > /dev/null 2> $OUT/tiled_fetch.err || fail "score_tiled under --pmc FETCH_SIZE"
python3 tools/prof_summary.py $OUT/tiled_fetch > $OUT/score_tiled_fetch.txt; cat $OUT/score_tiled_fetch.txt; find $OUT/tiled_fetch -name "*.csv" -delete

step "the prepared patches, on this copy of the tree"
git apply tools/patches/score_body_v2.patch || fail "score_body_v2.patch does not apply"
git apply tools/patches/topn_select_strided.patch || fail "topn_select_strided.patch does not apply"
python3 -c "import __graft_entry__ as g; g.build()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; fail "build with the patches"; }
timeout -k 10 90 python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; rc=$?; tail -2 $OUT/smoke.log | cut -c1-200; [ $rc = 0 ] || fail "smoke with the patches (rc $rc)"
timeout -k 10 420 python3 -m pytest tests/test_rm2_gpu.py tests/test_pruned_coop_gpu.py tests/test_multirank_gpu.py tests/test_random_small_gpu.py -m gpu -x -q > $OUT/tests_quick.log 2>&1; rc=$?
tail -6 $OUT/tests_quick.log; [ $rc = 0 ] || fail "tests with the patches (rc $rc): no bench"

step "bench with the patches"
timeout -k 10 150 python3 bench.py > $OUT/bench_patched.json 2> $OUT/bench_patched.err || fail "bench with the patches"
python3 - "$OUT/bench_patched.json" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("cold ms per job", round(d["ms_per_step"], 2), " phases", {k: round(v, 2) for k, v in d["phase_ms_rank0"].items()})
print("regime ms", {k: round(v["ms_per_step"], 1) for k, v in d.get("reference_regime", {}).items() if "ms_per_step" in v})
PY

step "the precision gate with the patches: every row of the full-size jobs against the fp64 definition, pruned = full pass"
timeout -k 10 300 python3 -m pytest tests/test_full_size_gpu.py -m gpu -x -q -k "fp64_definition or pruned_equals_full_pass or panel_mode" > $OUT/tests_fullsize.log 2>&1; rc=$?
tail -6 $OUT/tests_fullsize.log; [ $rc = 0 ] || fail "full-size parity with the patches (rc $rc)"
echo "ALL STEPS PASSED"
