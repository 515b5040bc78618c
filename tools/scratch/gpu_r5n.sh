#!/bin/bash
# front-end A/B in one call: the committed kernel, the pre-rotation select + fence (2 waves / SIMD), the same at 3 waves / SIMD
R=${GRAFT_REPO_ROOT:-/root/repo}
run() { echo -n "$1 "; LELE_HIP_LIBRARY=$2 timeout 200 python bench.py --no-model --no-yolo --no-cpu-baseline --steps 100 --warmup 10 2>&1 | tail -1 | grep -o '"value": [0-9.]*, \|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for i in 1 2; do
run orig ab_fe_orig.so
run B_2waves ab_fe_B.so
run A_3waves liblele_hip.so
done
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_fullsize_properties.py -m gpu -q -x 2>&1 | tail -3
