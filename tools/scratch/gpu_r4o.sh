#!/bin/bash
# round 4, call O: the persistent window kernel against the one-tile-per-workgroup one
mkdir -p gpurun_out/r4o
timeout 900 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4o/tests.log
echo rc=$?
timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4o/persist.json > gpurun_out/r4o/persist.log 2>&1 || tail -5 gpurun_out/r4o/persist.log
LELE_HIP_CONV_PERSIST=0 timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4o/tile.json > gpurun_out/r4o/tile.log 2>&1
python tools/conv_ab.py --compare gpurun_out/r4o/tile.json gpurun_out/r4o/persist.json
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4o/yolo_n64.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
LELE_HIP_CONV_PERSIST=0 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4o/yolo_n64_tile.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
