#!/bin/bash
export LELE_HIP_LAB=1
timeout 300 python tools/conv_ab.py --only "6 k3 s1\|8 k3 s1" --out gpurun_out/nar_new.json > /dev/null 2>&1
for sel in "->16 k3" "->8 k3"; do
timeout 300 python tools/conv_ab.py --only "$sel" --out gpurun_out/nar_new.json | grep "geom"
LELE_HIP_CONV_WIN_NARROW_MINC=100000 timeout 300 python tools/conv_ab.py --only "$sel" --out gpurun_out/nar_old.json | grep "geom"
done
for i in 1 2; do
echo -n "new "; timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "old "; LELE_HIP_CONV_WIN_NARROW_MINC=100000 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
