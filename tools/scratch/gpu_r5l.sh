#!/bin/bash
export LELE_HIP_LAB=1
timeout 300 python tools/conv_ab.py --only "s1 @80" --out gpurun_out/t_def.json > /dev/null 2>&1
LELE_HIP_CONV_TILE=40,6 timeout 300 python tools/conv_ab.py --only "s1 @80" --out gpurun_out/t_40x6.json > /dev/null 2>&1
LELE_HIP_CONV_TILE=80,3 timeout 300 python tools/conv_ab.py --only "s1 @80" --out gpurun_out/t_80x3.json > /dev/null 2>&1
LELE_HIP_CONV_TILE=20,12 timeout 300 python tools/conv_ab.py --only "s1 @80" --out gpurun_out/t_20x12.json > /dev/null 2>&1
python tools/conv_ab.py --compare gpurun_out/t_def.json gpurun_out/t_40x6.json
python tools/conv_ab.py --compare gpurun_out/t_def.json gpurun_out/t_80x3.json | tail -1
python tools/conv_ab.py --compare gpurun_out/t_def.json gpurun_out/t_20x12.json
