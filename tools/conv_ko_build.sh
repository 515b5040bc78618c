#!/bin/bash
# Builds the knock-out variants of the developer's library HERE (no GPU): lele_amd/liblele_hip_ko<bits>.so = the lab objects with
# conv.hip recompiled under -DLELE_CONV_KO=<bits>.  Loaded with LELE_HIP_LIBRARY=liblele_hip_ko<bits>.so (tools/conv_ko.sh).
cd "$(dirname "$0")/.."
LELE_HIP_LAB=1 python -m lele_amd.build > /dev/null
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-function -Wno-unused-variable -Wno-unused-result -DLELE_HIP_LAB=1"
for ko in "$@"; do
  ( hipcc $F -DLELE_CONV_KO=$ko -c lele_amd/csrc/conv.hip -o lele_amd/_build_lab/conv_ko$ko.o 2>/dev/null &&
    hipcc -shared -fPIC --offload-arch=gfx950 -o lele_amd/liblele_hip_ko$ko.so $(ls lele_amd/_build_lab/*.o | grep -v "conv") lele_amd/_build_lab/conv_ko$ko.o -ldl && echo built ko$ko ) &
done
wait
