#!/bin/bash
# convolution epilogues: v_exp_f32 / v_rcp_f32 SiLU against the replica of the reference's polynomial (whole forward, interleaved)
for i in 1 2 3; do
for l in ab_silu_exact.so liblele_hip.so; do echo -n "$l "; LELE_HIP_LIBRARY=$l timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'; done
done
LELE_HIP_LIBRARY=liblele_hip.so timeout 600 python -m pytest tests/test_conv_rnn.py tests/test_channel_views.py tests/test_fullsize_properties.py -m gpu -q 2>&1 | tail -3
