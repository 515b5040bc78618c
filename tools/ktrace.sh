#!/bin/bash
# Runs ON THE GPU BOX: per-DISPATCH kernel trace (rocprofv3 --kernel-trace, csv) of an arbitrary command, for tools/ktrace_graph.py.
# usage: tools/ktrace.sh <tag> <command...>      -> gpurun_out/ktrace_<tag>/k_kernel_trace.csv
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/ktrace_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 600 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o k -- "$@" > "$OUT/cmd.log" 2>&1 ) || tail -5 "$OUT/cmd.log"
ls -la "$OUT" | head
