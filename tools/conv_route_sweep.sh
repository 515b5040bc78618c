#!/bin/bash
# Runs ON THE GPU BOX (lab build): the reference Yolo graph (linear) under the window kernels' routing switches, one value at a time.
export LELE_HIP_LAB=1
run() { python tools/yolo_lifted_batch.py --batch 64 --check 1 --runs 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', d['graph_ms_per_forward'])"; }
run base; run base
for v in 0 2 8; do LELE_HIP_CONV_TILE_BETA=$v run "TILE_BETA=$v"; done
for v in 16 48 64; do LELE_HIP_CONV_W1_MINC=$v run "W1_MINC=$v"; done
for v in 128 512; do LELE_HIP_CONV_W1_MAXOC=$v run "W1_MAXOC=$v"; done
for v in 100 1600; do LELE_HIP_CONV_W1_MINPLANE=$v run "W1_MINPLANE=$v"; done
for v in 16 64; do LELE_HIP_CONV_WIN_NARROW_MINC=$v run "WIN_NARROW_MINC=$v"; done
LELE_HIP_CONV_OCT128=0 run "OCT128=0"
run base
