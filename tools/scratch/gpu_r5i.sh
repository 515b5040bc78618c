#!/bin/bash
timeout 900 python -m pytest tests/test_conv_rnn.py tests/test_fullsize_properties.py -m gpu -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -5
export LELE_HIP_LAB=1
for i in 1 2; do
echo -n "window "; timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
echo -n "gemm   "; LELE_HIP_CONVT_GEMM=1 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done
