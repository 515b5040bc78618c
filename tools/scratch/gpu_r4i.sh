#!/bin/bash
# round 4, call I: runtime tile shapes of the window kernels
mkdir -p gpurun_out/r4i
timeout 900 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4i/tests.log
echo rc=$?
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --table gpurun_out/r4i/yolo_table.json --out gpurun_out/r4i/yolo_n64.json 2>&1 | tail -3 | cut -c1-1500 | tee gpurun_out/r4i/yolo.log
LELE_HIP_CONV_TILE=32,8 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4i/yolo_n64_tile32x8.json 2>&1 | tail -1 | cut -c1-200
timeout 400 python tools/yolo_lifted_batch.py --batch 64 --out gpurun_out/r4i/yolo26seg_lifted_n64.json 2>&1 | tail -1 | cut -c1-1500
