#!/bin/bash
export LELE_HIP_LAB=1
for i in 1 2; do
echo -n "maxoc256 plane400        "; LELE_HIP_CONV_W1_MAXOC=256 LELE_HIP_CONV_W1_MINPLANE=400 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "maxoc512 plane400        "; LELE_HIP_CONV_W1_MAXOC=512 LELE_HIP_CONV_W1_MINPLANE=400 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "maxoc256 plane400 minc16 "; LELE_HIP_CONV_W1_MAXOC=256 LELE_HIP_CONV_W1_MINPLANE=400 LELE_HIP_CONV_W1_MINC=16 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "maxoc256 plane100        "; LELE_HIP_CONV_W1_MAXOC=256 LELE_HIP_CONV_W1_MINPLANE=100 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
LELE_HIP_CONV_W1_MAXOC=256 LELE_HIP_CONV_W1_MINPLANE=400 timeout 400 python tools/yolo_lifted_batch.py --batch 64 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
timeout 400 python tools/yolo_lifted_batch.py --batch 64 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
