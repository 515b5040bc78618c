#!/bin/bash
# Runs ON THE GPU BOX: the window kernels' time with parts knocked out at COMPILE time (tools/conv_ko_build.sh builds
# lele_amd/liblele_hip_ko<bits>.so: 2 stores, 4 five of six products, 8 window loads), per geometry filter.
# usage: tools/conv_ko.sh "<filter>" ...     (a runtime switch in the product loop distorts it: 3-5 x slower kernels)
export LELE_HIP_LAB=1
for f in "$@"; do
  python tools/conv_ab.py --only "$f" 2>/dev/null | grep geom | sed "s/^/ko=0 /"
  for ko in ${KOS:-2 4 8 14}; do
    LELE_HIP_LIBRARY=liblele_hip_ko$ko.so python tools/conv_ab.py --only "$f" 2>/dev/null | grep geom | sed "s/^/ko=$ko /"
  done
done
