#!/bin/bash
# round 4, call K: conv2d_res (residual adds fused), under-filled tile rule
mkdir -p gpurun_out/r4k
timeout 900 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py tests/test_lift_generated.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4k/tests.log
echo rc=$?
timeout 300 python tools/conv_ab.py --out gpurun_out/r4k/pick.json > gpurun_out/r4k/pick.log 2>&1 || tail -5 gpurun_out/r4k/pick.log
LELE_HIP_CONV_TILE=rows timeout 300 python tools/conv_ab.py --out gpurun_out/r4k/rows.json > gpurun_out/r4k/rows.log 2>&1
python tools/conv_ab.py --compare gpurun_out/r4k/rows.json gpurun_out/r4k/pick.json | grep -E "@20|sum"
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --table gpurun_out/r4k/yolo_table.json --out gpurun_out/r4k/yolo_n64.json 2>&1 | tail -1 | cut -c1-1600 | tee gpurun_out/r4k/yolo.log
LELE_HIP_CONV_TILE=rows timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4k/yolo_n64_rows.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
timeout 400 python tools/yolo_lifted_batch.py --batch 64 --out gpurun_out/r4k/yolo26seg_lifted_n64.json 2>&1 | tail -1 | cut -c1-1500
