#!/bin/bash
# round 4, call S: loads of 3-4 chunks in flight (new) against 2 (liblele_hip_prev.so), whole forward interleaved
mkdir -p gpurun_out/r4s
timeout 600 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py tests/test_eltwise_norm.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r4s/tests.log
for i in 1 2 3; do
echo -n "new  "; timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4s/yolo_new$i.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "prev "; LELE_HIP_LIBRARY=liblele_hip_prev.so timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4s/yolo_prev$i.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --table gpurun_out/r4s/yolo_table_new.json --out gpurun_out/r4s/yolo_new.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
LELE_HIP_LIBRARY=liblele_hip_prev.so timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --table gpurun_out/r4s/yolo_table_prev.json --out gpurun_out/r4s/yolo_prev.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
