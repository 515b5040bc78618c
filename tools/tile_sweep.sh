mkdir -p gpurun_out/r6s1; export LELE_HIP_LAB=1
run() { # geom-filter, tile
  LELE_HIP_CONV_TILE=$2 python tools/conv_ab.py --only "$1" 2>/dev/null | grep geom | sed "s/^/tile=$2 /"
}
for t in "" 20,12 20,6 20,5 20,4 20,3; do run "k3 s1 @20" "$t"; done
for t in "" 40,6 40,5 40,3 20,10 20,5 40,2; do run "32->32 k3 s1 @40" "$t"; run "64->64 k3 s1 @40" "$t"; done
for t in "" 20,6 20,5 20,3 20,2; do run "k3 s2 @20" "$t"; done
for t in "" 40,3 40,2 20,4 20,2 40,1; do run "64->64 k3 s2 @40" "$t"; done
