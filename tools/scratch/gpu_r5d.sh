#!/bin/bash
mkdir -p gpurun_out/r5d
timeout 900 python -m pytest tests/test_channel_views.py tests/test_lift_generated.py tests/test_conv_rnn.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
echo -n "lifted fold  "; timeout 400 python tools/yolo_lifted_batch.py --batch 64 --out gpurun_out/r5d/lifted_new.json 2>&1 | tail -1 | grep -o '"transposed_splits_folded": [0-9]*\|"graph_ms_per_forward": [0-9.]*\|"max_error_in_units_of_1e-4_per_output": [^]]*]' | tr '\n' ' '; echo
echo -n "lifted three "; LELE_AMD_FOLD_TAILS=0 timeout 400 python tools/yolo_lifted_batch.py --batch 64 --out gpurun_out/r5d/lifted_prev.json 2>&1 | tail -1 | grep -o '"transposed_splits_folded": [0-9]*\|"graph_ms_per_forward": [0-9.]*' | tr '\n' ' '; echo
done
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r5d/yolo.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
