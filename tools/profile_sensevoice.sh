#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): per-kernel time table of the SenseVoice-shaped encoder (eager, one config per run).
# usage: tools/profile_sensevoice.sh <tag>       (outputs under gpurun_out/prof_sv_<tag>/{c3,c4}[_compiled]_kernel_stats.csv)
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_sv_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
LEGS=${LEGS:-both}   # LEGS=compiled: only the compiled-plan leg
for C in c3 c4; do
  [ "$LEGS" = compiled ] || timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o $C -- \
      python $R/tools/sensevoice_graph.py --no-graph --configs $C --runs 4 > "$OUT/$C.json" 2> "$OUT/$C.log"
  # the same model compiled from ONNX (lele_amd.compiler, all fused forms), 10 eager forwards
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${C}_compiled -- \
      python $R/tools/sensevoice_graph.py --compiled-only --configs $C --runs 8 > "$OUT/${C}_compiled.json" 2> "$OUT/${C}_compiled.log"
done
find "$OUT" -name '*kernel_stats.csv'
