#!/bin/bash
# round 4, call U: ConvInteger on the i8 matrix cores (equality with the oracle), the bounds-asserting build, job-token rendezvous, long-row min/max
mkdir -p gpurun_out/r4u
timeout 900 python -m pytest tests/test_conv_integer.py tests/test_comm_gpu.py tests/test_eltwise_norm.py tests/test_quant.py -m gpu -x -q 2>&1 | tail -12 | tee gpurun_out/r4u/tests.log
timeout 1500 python -m pytest tests/test_debug_bounds.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r4u/tests_dbg.log
LELE_HIP_LAB=1 timeout 300 python tools/conv_integer_bench.py | tee gpurun_out/r4u/conv_integer_i8.json
LELE_HIP_LAB=1 LELE_HIP_CONV_INTEGER_F32=1 timeout 300 python tools/conv_integer_bench.py | tee gpurun_out/r4u/conv_integer_f32.json
