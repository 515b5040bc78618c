#!/bin/bash
# round 4, call X: three forms of the persistent kernel's epilogue (A: as committed; B: lane-derived values recomputed; C: B + strip by strip)
mkdir -p gpurun_out/r4x
for v in A B C; do
LELE_HIP_LIBRARY=liblele_hip_$v.so timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4x/$v.json > gpurun_out/r4x/$v.log 2>&1 || tail -3 gpurun_out/r4x/$v.log
done
python tools/conv_ab.py --compare gpurun_out/r4x/A.json gpurun_out/r4x/B.json
python tools/conv_ab.py --compare gpurun_out/r4x/A.json gpurun_out/r4x/C.json | tail -1
for i in 1 2; do for v in A B C; do
echo -n "$v "; LELE_HIP_LIBRARY=liblele_hip_$v.so timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --out gpurun_out/r4x/yolo_$v$i.json 2>&1 | tail -1 | grep -o '"graph_ms_per_forward": [0-9.]*'
done; done
