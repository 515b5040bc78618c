#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 5's measurements -> gpurun_out/prof_r05/ ; condensed into profiles/r05_* by hand-off below.
# Counters are not collected this round (fe_main_kernel is unchanged: profiles/frontend_roofline.json stays round 4's); every rocprofv3
# pass here is --kernel-trace (+ --stats) only.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r05
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 1. the default bench invocation, plain; the front-end leg under rocprofv3 --kernel-trace --stats
timeout 400 python $R/bench.py > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python $R/bench.py --no-model --no-yolo --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
# 2. per-kernel tables of the compiled SenseVoice-shaped plan (configs[2] and one configs[3] shard)
for C in c3 c4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sv" -o ${C}_compiled -- \
      python $R/tools/sensevoice_graph.py --compiled-only --configs $C --runs 8 > "$OUT/sv_${C}.json" 2> "$OUT/sv_${C}.log"
done
# 3. configs[4]: the reference's generated graph re-batched (table with bounds), and its kernel table under rocprofv3
timeout 300 python $R/tools/yolo_lifted_batch.py --batch 64 --check 4 --table "$OUT/yolo_lifted_table.json" --out "$OUT/yolo_lifted_n64.json" > "$OUT/yolo_lifted.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/yolo" -o lifted -- \
    python $R/tools/yolo_lifted_batch.py --batch 64 --check 0 --runs 5 > "$OUT/yolo_lifted_prof.log" 2>&1
timeout 300 python $R/tools/yolo_graph.py --batch 64 --check 4 --out "$OUT/yolo_n64.json" > "$OUT/yolo.log" 2>&1
# 4. linear graph vs DAG: timings, and a kernel trace of the DAG replays (do kernels of different branches overlap in time?)
timeout 300 python $R/tools/dag_bench.py --lanes 3 --gain 0.04 --out "$OUT/dag_bench.json" > "$OUT/dag_bench.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/dagtrace" -o lifted -- \
    python $R/tools/dag_bench.py --only lifted --lanes 3 --gain 0.04 --runs 3 > "$OUT/dag_trace.log" 2>&1
python3 $R/tools/trace_overlap.py "$OUT/dagtrace" > "$OUT/dag_trace_overlap.json" 2>> "$OUT/dag_trace.log"
# 5. where a graph-level difference comes from (after the rounded pieces), and the matrix core's rounding probe
timeout 300 python $R/tools/graph_error_growth.py --batch 64 --local --out "$OUT/graph_error_growth_after.json" > "$OUT/growth.log" 2>&1
[ -x $R/tools/mfma_round ] && $R/tools/mfma_round > "$OUT/mfma_round.json" 2> "$OUT/mfma_round.err"
# 6. operator micro-benchmarks (the convolution / attention rows changed: rounded pieces)
timeout 280 python $R/tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.log" 2>&1
timeout 120 python $R/tools/attention_bench.py > "$OUT/attention_bench.json" 2> "$OUT/attention_bench.log"
rm -rf "$OUT"/dagtrace/*/*.db 2>/dev/null
find "$OUT" -name '*.csv' | wc -l
du -sh "$OUT"
tail -c 300 "$OUT/bench_plain.json"
