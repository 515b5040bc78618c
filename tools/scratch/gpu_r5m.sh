#!/bin/bash
export LELE_HIP_LAB=1
timeout 300 python tools/conv_ab.py --out gpurun_out/b4.json > /dev/null 2>&1
LELE_HIP_CONV_TILE_BETA=0 timeout 300 python tools/conv_ab.py --out gpurun_out/b0.json > /dev/null 2>&1
LELE_HIP_CONV_TILE_BETA=8 timeout 300 python tools/conv_ab.py --out gpurun_out/b8.json > /dev/null 2>&1
python tools/conv_ab.py --compare gpurun_out/b0.json gpurun_out/b4.json | grep -v " +0\.[0-4] %\| -0\.[0-4] %"
python tools/conv_ab.py --compare gpurun_out/b0.json gpurun_out/b8.json | tail -1
for i in 1 2; do
echo -n "beta4 "; timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "beta0 "; LELE_HIP_CONV_TILE_BETA=0 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
