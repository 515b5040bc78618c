#!/bin/bash
# usage (on the GPU box): tools/pmc_ops.sh <what: quant|matmul|conv> <tag>   -> gpurun_out/pmc_<tag>/<counter set>/
set -u
WHAT=${1:-matmul}; TAG=${2:-x}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VMEM SQ_INSTS_FLAT" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "FETCH_SIZE WRITE_SIZE TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum TA_BUSY_avr TA_TA_BUSY_sum" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/set$i" -o ops -- ${PMC_CMD:-python $R/tools/prof_ops.py $WHAT} > "$OUT/set$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/set*/ops_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
import os
flt = os.environ.get("PMC_FILTER", "")
for k, d in agg.items():
    if flt and flt not in k: continue
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s n=%d avg=%.4g" % (c, len(v), sum(v) / len(v)))
PY
