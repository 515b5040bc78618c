#!/bin/bash
# round 4, call Y: the stride-2 window kernel as persistent workgroups (new) against one workgroup per tile (liblele_hip_prev.so)
mkdir -p gpurun_out/r4y
timeout 900 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4y/tests.log
timeout 300 python tools/conv_ab.py --only "s2 @" --out gpurun_out/r4y/new.json > gpurun_out/r4y/new.log 2>&1 || tail -3 gpurun_out/r4y/new.log
LELE_HIP_LIBRARY=liblele_hip_prev.so timeout 300 python tools/conv_ab.py --only "s2 @" --out gpurun_out/r4y/prev.json > gpurun_out/r4y/prev.log 2>&1
python tools/conv_ab.py --compare gpurun_out/r4y/prev.json gpurun_out/r4y/new.json
for i in 1 2; do
echo -n "new  "; timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4y/yolo_new$i.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "prev "; LELE_HIP_LIBRARY=liblele_hip_prev.so timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4y/yolo_prev$i.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
