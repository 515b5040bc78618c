#!/bin/bash
# front-end: the bare v_log_f32 against __logf (same bits?), timing in the same call
for l in ab_fe_prev.so liblele_hip.so; do echo -n "$l "; PYTHONPATH=. LELE_HIP_LIBRARY=$l python tools/scratch/fe_hash.py 2>&1 | tail -1; done
run() { echo -n "$1 "; LELE_HIP_LIBRARY=$1 timeout 200 python bench.py --no-model --no-yolo --no-cpu-baseline --steps 100 --warmup 10 2>&1 | tail -1 | grep -o '"value": [0-9.]*, \|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for i in 1 2; do run ab_fe_prev.so; run liblele_hip.so; done
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_fullsize_properties.py -m gpu -q -x 2>&1 | tail -3
