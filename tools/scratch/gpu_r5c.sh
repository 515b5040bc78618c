#!/bin/bash
export LELE_HIP_LAB=1
for kb in 64 32 24 16; do
LELE_HIP_DW_LDS_KB=$kb timeout 300 python tools/conv_ab.py --dw --out gpurun_out/dw_$kb.json > gpurun_out/dw_$kb.log 2>&1 || tail -3 gpurun_out/dw_$kb.log
done
for kb in 32 24 16; do python tools/conv_ab.py --compare gpurun_out/dw_64.json gpurun_out/dw_$kb.json; done
