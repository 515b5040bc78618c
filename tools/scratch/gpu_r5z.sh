#!/bin/bash
# conv epilogue SiLU: 5-instruction form against the form with the product's rounding compensated (time; batch-64 against batch-1 rows)
for i in 1 2; do
for l in ab_silu5.so liblele_hip.so; do echo -n "$l "; LELE_HIP_LIBRARY=$l timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'; done
done
for l in ab_silu5.so liblele_hip.so; do echo "== $l"; LELE_HIP_LIBRARY=$l timeout 600 python -m pytest tests/test_lift_generated.py tests/test_conv_rnn.py -m gpu -q -k "batch_64_as_one_graph or epilogue_silu" 2>&1 | grep -E "^E +Assert|passed|failed" | head; 
LELE_HIP_LIBRARY=$l timeout 300 python tools/yolo_lifted_batch.py --batch 64 2>&1 | tail -1 | grep -o '"max_error_in_units_of_1e-4_per_output": [^]]*]\|"detection_rows_in_a_different_order": [0-9]*\|"graph_ms_per_forward": [0-9.]*' | tr '\n' ' '; echo; done
