#!/bin/bash
# round 4, call M: does staggering the two workgroups of a CU overlap one's loads with the other's stores?
mkdir -p gpurun_out/r4m
export LELE_HIP_LAB=1
for ko in 0 16 32 64 128; do
LELE_HIP_CONV_KO=$ko timeout 300 python tools/conv_ab.py --only "s1 @" --out gpurun_out/r4m/ko_$ko.json > gpurun_out/r4m/ko_$ko.log 2>&1 || tail -3 gpurun_out/r4m/ko_$ko.log
done
python - <<'PY'
import json
K=(0,16,32,64,128)
R={k:json.load(open('gpurun_out/r4m/ko_%d.json'%k))['rows'] for k in K}
print("%-24s"%"geometry"+"".join("%9s"%("st%d"%(k>>4)) for k in R))
for i,r in enumerate(R[0]):
    print("%-24s"%r['geom']+"".join("%9.1f"%R[k][i]['us'] for k in R))
PY
