#!/bin/bash
# front-end: 48-frame workgroups at four waves per SIMD against 64-frame ones at three (lab build, same source), and the round's
# committed kernel (saved library)
run() { echo -n "$1 passes=$2 "; LELE_HIP_LAB=$3 LELE_HIP_FE_PASSES=$2 LELE_HIP_LIBRARY=$1 timeout 200 python bench.py --no-model --no-yolo --no-cpu-baseline --steps 100 --warmup 10 2>&1 | tail -1 | grep -o '"value": [0-9.]*, \|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for i in 1 2; do
run ab_fe_orig.so 4 0
run liblele_hip_lab.so 4 1
run liblele_hip_lab.so 3 1
run liblele_hip.so 3 0
done
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_fullsize_properties.py tests/test_fullsize_graph.py -m gpu -q -x 2>&1 | tail -3
