#!/bin/bash
# Runs ON THE GPU BOX: the reference Yolo graph (linear) with two builds of the library, interleaved.   usage: tools/ab_lib.sh <libA.so> <libB.so>
run() { LELE_HIP_LIBRARY=$1 python tools/yolo_lifted_batch.py --batch 64 --check 1 --runs 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['graph_ms_per_forward'])"; }
for i in 1 2 3; do run $1; run $2; done
