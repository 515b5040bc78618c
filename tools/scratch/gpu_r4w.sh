#!/bin/bash
# round 4, call W: the batch attention kernel with a per-workgroup start tile
mkdir -p gpurun_out/r4w
export LELE_HIP_LAB=1
for rot in 0 1 5 7 0 1; do
echo -n "rot=$rot "; LELE_HIP_ATTN_ROT=$rot timeout 200 python tools/attention_bench.py --only default --reps 60 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin)
print({k:list(v.values())[0]['us'] for k,v in d.items()})"
done
