#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): counters of the configs[3]-shard kernels after round 5's i8 GEMM work -> gpurun_out/prof_r05pmc/.
# Three separate --pmc passes (kernel-trace only beside them, as the pool requires): SQ busy / instruction counters, FETCH_SIZE, WRITE_SIZE.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r05pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/sensevoice_graph.py --compiled-only --configs c4 --runs 3 --layers 10"
timeout 250 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d "$OUT/sq" -o c4 -- $CMD > "$OUT/sq.json" 2> "$OUT/sq.log"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 250 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$C" -o c4 -- $CMD > "$OUT/$C.json" 2> "$OUT/$C.log"
done
find "$OUT" -name '*counter_collection.csv' | xargs ls -la
