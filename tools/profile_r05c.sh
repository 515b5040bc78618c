#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): round 5, after the XCD-contiguous item order of the convolution kernels -> gpurun_out/prof_r05c/ :
# the reference graph's statement table and kernel table again, and the linear-vs-DAG bench.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r05c
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 python $R/tools/yolo_lifted_batch.py --batch 64 --check 4 --table "$OUT/yolo_lifted_table.json" --out "$OUT/yolo_lifted_n64.json" > "$OUT/yolo_lifted.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/yolo" -o lifted -- \
    python $R/tools/yolo_lifted_batch.py --batch 64 --check 0 --runs 5 > "$OUT/yolo_lifted_prof.log" 2>&1
timeout 300 python $R/tools/yolo_graph.py --batch 64 --check 4 --out "$OUT/yolo_n64.json" > "$OUT/yolo.log" 2>&1
timeout 300 python $R/tools/dag_bench.py --lanes 3 --gain 0.04 --out "$OUT/dag_bench.json" > "$OUT/dag_bench.log" 2>&1
find "$OUT" -name '*kernel_stats.csv'
tail -c 400 "$OUT/yolo_lifted_n64.json"
