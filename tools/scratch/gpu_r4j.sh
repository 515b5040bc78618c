#!/bin/bash
# round 4, call J: tile shapes A/B inside one call
mkdir -p gpurun_out/r4j
for i in 1 2; do
timeout 300 python tools/conv_ab.py --out gpurun_out/r4j/pick_$i.json > gpurun_out/r4j/pick_$i.log 2>&1 || tail -5 gpurun_out/r4j/pick_$i.log
LELE_HIP_CONV_TILE=rows timeout 300 python tools/conv_ab.py --out gpurun_out/r4j/rows_$i.json > gpurun_out/r4j/rows_$i.log 2>&1
done
python tools/conv_ab.py --compare gpurun_out/r4j/rows_1.json gpurun_out/r4j/pick_1.json
python tools/conv_ab.py --compare gpurun_out/r4j/rows_2.json gpurun_out/r4j/pick_2.json | tail -1
python tools/conv_ab.py --compare gpurun_out/r4j/rows_1.json gpurun_out/r4j/rows_2.json | tail -1
