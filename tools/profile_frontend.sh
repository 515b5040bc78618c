#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): collects the rocprofv3 evidence behind bench.py's roofline numbers.
#   1. --kernel-trace --stats of the exact bench command  -> per-kernel average durations
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE)      -> HBM bytes per fe_main_kernel launch
# Counters are collected in their own runs with --kernel-trace only (no sys/hip/hsa trace domains).
# usage: tools/profile_frontend.sh <tag>        (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py"   # the default invocation (200 steps after 20 warm-up steps: the clocks have settled)
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- $BENCH > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o bench -- $BENCH --no-cpu-baseline > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.log"
done
timeout 300 $BENCH > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"
find "$OUT" -name '*.csv' | head -50
tail -c 600 "$OUT/bench_plain.json"
