#!/bin/bash
# round 4, call Z: the detection tail as transposing copies
mkdir -p gpurun_out/r4z
timeout 900 python -m pytest tests/test_channel_views.py tests/test_lift_generated.py tests/test_fullsize_properties.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r4z/tests.log
for i in 1 2; do
echo -n "fold  "; timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4z/yolo_new$i.json 2>&1 | tail -1 | grep -o '"transposed_splits_folded": [0-9]*\|"graph_ms_per_forward": [0-9.]*' | tr '\n' ' '; echo
echo -n "three "; LELE_AMD_FOLD_TAILS=0 timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4z/yolo_prev$i.json 2>&1 | tail -1 | grep -o '"transposed_splits_folded": [0-9]*\|"graph_ms_per_forward": [0-9.]*' | tr '\n' ' '; echo
done
echo -n "lifted fold  "; timeout 400 python tools/yolo_lifted_batch.py --batch 64 --out gpurun_out/r4z/lifted_new.json 2>&1 | tail -1 | grep -o '"transposed_splits_folded": [0-9]*\|"graph_ms_per_forward": [0-9.]*' | tr '\n' ' '; echo
echo -n "lifted three "; LELE_AMD_FOLD_TAILS=0 timeout 400 python tools/yolo_lifted_batch.py --batch 64 --out gpurun_out/r4z/lifted_prev.json 2>&1 | tail -1 | grep -o '"transposed_splits_folded": [0-9]*\|"graph_ms_per_forward": [0-9.]*' | tr '\n' ' '; echo
