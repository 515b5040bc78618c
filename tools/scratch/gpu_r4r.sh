#!/bin/bash
# round 4, call R: exhaustive check of the short reciprocal; top-k with wave scans; whole-forward timing
mkdir -p gpurun_out/r4r
timeout 300 tools/recip_check | tee gpurun_out/r4r/recip_check.json
timeout 900 python -m pytest tests/test_manip.py tests/test_eltwise_norm.py tests/test_conv_rnn.py tests/test_lift_generated.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4r/tests.log
for i in 1 2; do
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --table gpurun_out/r4r/yolo_table.json --out gpurun_out/r4r/yolo_n64.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
timeout 400 python tools/yolo_lifted_batch.py --batch 64 --out gpurun_out/r4r/yolo26seg_lifted_n64.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
