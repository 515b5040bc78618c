#!/bin/bash
# round 4, call N: what the SiLU costs (the same convolutions with and without the activation)
mkdir -p gpurun_out/r4n
timeout 300 python tools/conv_ab.py --out gpurun_out/r4n/silu.json > gpurun_out/r4n/silu.log 2>&1 || tail -3 gpurun_out/r4n/silu.log
timeout 300 python tools/conv_ab.py --act none --out gpurun_out/r4n/none.json > gpurun_out/r4n/none.log 2>&1
timeout 300 python tools/conv_ab.py --act relu --out gpurun_out/r4n/relu.json > gpurun_out/r4n/relu.log 2>&1
python tools/conv_ab.py --compare gpurun_out/r4n/silu.json gpurun_out/r4n/none.json
python tools/conv_ab.py --compare gpurun_out/r4n/silu.json gpurun_out/r4n/relu.json | tail -1
