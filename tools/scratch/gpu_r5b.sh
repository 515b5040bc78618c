#!/bin/bash
mkdir -p gpurun_out/r5b
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r5b/tests.log
for i in 1 2; do timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --table gpurun_out/r5b/yolo_table.json --out gpurun_out/r5b/yolo.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'; done
