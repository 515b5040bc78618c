#!/bin/bash
# usage (on the GPU box): tools/pmc_frontend.sh <tag>  -> gpurun_out/pmcfe_<tag>/ : SQ/GRBM counters of fe_main_kernel
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmcfe_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for SET in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-trace --output-format csv -d "$OUT/set$i" -o fe -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > "$OUT/set$i.log" 2>&1
done
python3 - "$OUT" <<'PY'
import csv, sys, glob, collections, json
out = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob(out + "/set*/fe_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "fe_main_kernel" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
