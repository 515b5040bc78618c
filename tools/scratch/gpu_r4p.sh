#!/bin/bash
# round 4, call P: persistent window kernel with the blocks of an under-filled layer spread over workgroups; 32-channel blocks everywhere?
mkdir -p gpurun_out/r4p
timeout 900 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4p/tests.log
timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4p/persist.json > gpurun_out/r4p/persist.log 2>&1 || tail -5 gpurun_out/r4p/persist.log
LELE_HIP_CONV_PERSIST=0 timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4p/tile.json > gpurun_out/r4p/tile.log 2>&1
LELE_HIP_CONV_OCT=32 timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4p/oct32.json > gpurun_out/r4p/oct32.log 2>&1
python tools/conv_ab.py --compare gpurun_out/r4p/tile.json gpurun_out/r4p/persist.json
python tools/conv_ab.py --compare gpurun_out/r4p/persist.json gpurun_out/r4p/oct32.json
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4p/yolo_n64.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
LELE_HIP_CONV_OCT=32 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4p/yolo_n64_oct32.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
