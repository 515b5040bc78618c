#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 5, second half (after the i8 GEMM changes) -> gpurun_out/prof_r05b/ ; condensed into profiles/r05_*.
# Only what those changes touch: the default bench line, the SenseVoice-shaped kernel tables, the quantised-linear micro-benchmarks and the
# reference graph's statement table (top-k).  The front-end, DAG, error-growth and attention profiles of profile_r05.sh stay valid.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r05b
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"
for C in c3 c4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sv" -o ${C}_compiled -- \
      python $R/tools/sensevoice_graph.py --compiled-only --configs $C --runs 8 > "$OUT/sv_${C}.json" 2> "$OUT/sv_${C}.log"
done
timeout 300 python $R/tools/yolo_lifted_batch.py --batch 64 --check 4 --table "$OUT/yolo_lifted_table.json" --out "$OUT/yolo_lifted_n64.json" > "$OUT/yolo_lifted.log" 2>&1
timeout 280 python $R/tools/microbench.py --only quant,c4 --out "$OUT/microbench_quant.json" > "$OUT/microbench.log" 2>&1
find "$OUT" -name '*.csv' | wc -l
du -sh "$OUT"
tail -c 300 "$OUT/bench_plain.json"
