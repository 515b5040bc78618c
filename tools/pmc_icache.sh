#!/bin/bash
# Instruction-cache counters of one kernel (name substring) over a command -> per-launch averages
# usage: tools/pmc_icache.sh <kernel-substring> <tag> -- <command...>
K=$1; TAG=$2; shift 3
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/$TAG; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
i=0
for C in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
         "SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
         "SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -o k -- "$@" > "$OUT/p$i.log" 2>&1 || tail -3 "$OUT/p$i.log"
done
python - <<PY
import csv, glob, collections
v = collections.defaultdict(list)
for f in sorted(glob.glob("$OUT/p*/k_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if "$K" in r["Kernel_Name"]:
            v[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, x in sorted(v.items()):
    print("%-28s %14.0f  (n=%d)" % (k, sum(x) / len(x), len(x)))
PY
