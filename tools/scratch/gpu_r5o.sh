#!/bin/bash
# front-end A/B in one call: the committed kernel against the peeled-last-pass forms (DPP fold, load position, SLP on / off)
run() { echo -n "$1 "; LELE_HIP_LIBRARY=$1 timeout 200 python bench.py --no-model --no-yolo --no-cpu-baseline --steps 100 --warmup 10 2>&1 | tail -1 | grep -o '"value": [0-9.]*, \|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for i in 1 2; do
for l in ab_fe_orig.so ab_fe_P_s1_k2_p0_slp0.so ab_fe_P_s1_k2_p2_slp0.so ab_fe_P_s1_k2_p3_slp0.so ab_fe_P_s0_k0_p3_slp1.so; do run $l; done
done
LELE_HIP_LIBRARY=ab_fe_P_s1_k2_p3_slp0.so timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_fullsize_properties.py -m gpu -q -x 2>&1 | tail -3
