#!/bin/bash
# FSMN depthwise convolution: one channel per lane against four (16-byte accesses), 4 or 8 time steps per thread (lab build)
export LELE_HIP_LAB=1
for cfg in "0 8" "1 4" "1 8" "0 8" "1 4" "1 8"; do set -- $cfg; echo -n "quad=$1 tt=$2: "; LELE_HIP_TLC_QUAD=$1 LELE_HIP_TLC_TT=$2 python tools/scratch/tlc_bench.py 2>&1 | tail -1; done
