#!/bin/bash
# round 4, call Q: interleaved whole-forward A/B (persistent / one tile per workgroup)
mkdir -p gpurun_out/r4q
for i in 1 2 3; do
echo -n "persist "; timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4q/yolo_p$i.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "tile    "; LELE_HIP_CONV_PERSIST=0 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4q/yolo_t$i.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4q/persist.json > gpurun_out/r4q/persist.log 2>&1 || tail -5 gpurun_out/r4q/persist.log
LELE_HIP_CONV_PERSIST=0 timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4q/tile.json > gpurun_out/r4q/tile.log 2>&1
python tools/conv_ab.py --compare gpurun_out/r4q/tile.json gpurun_out/r4q/persist.json | grep -E "@20|@40|sum"
timeout 900 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py -m gpu -x -q 2>&1 | tail -3 | tee gpurun_out/r4q/tests.log
