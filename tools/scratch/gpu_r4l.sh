#!/bin/bash
# round 4, call L: what the 1 x 1 window kernel spends its time on (knock-outs in the developer's build)
mkdir -p gpurun_out/r4l
export LELE_HIP_LAB=1
for ko in 0 1 2 4 5 6 7 8; do
LELE_HIP_CONV_KO=$ko timeout 300 python tools/conv_ab.py --only "k1 " --out gpurun_out/r4l/ko_$ko.json > gpurun_out/r4l/ko_$ko.log 2>&1 || tail -3 gpurun_out/r4l/ko_$ko.log
done
python - <<'PY'
import json
R={k:json.load(open('gpurun_out/r4l/ko_%d.json'%k))['rows'] for k in (0,1,2,4,5,6,7,8)}
print("%-24s"%"geometry"+"".join("%9s"%("ko%d"%k) for k in R))
for i,r in enumerate(R[0]):
    print("%-24s"%r['geom']+"".join("%9.1f"%R[k][i]['us'] for k in R))
PY
