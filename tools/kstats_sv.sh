#!/bin/bash
# rocprofv3 kernel stats of the compiled SenseVoice-shaped encoder (eager forwards) for one config -> gpurun_out/<tag>/
# usage: tools/kstats_sv.sh <c3|c4> <tag>
C=${1:-c4}; TAG=${2:-kstats}
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o ${C}_compiled -- \
    python $R/tools/sensevoice_graph.py --compiled-only --configs $C --runs 8 > "$OUT/sv_${C}.json" 2> "$OUT/sv_${C}.log"
python - <<PY
import csv
rows=list(csv.DictReader(open("$OUT/${C}_compiled_kernel_stats.csv")))
tot=0
for r in rows:
    nm=r["Name"]
    if "wpack" in nm or "copyBuffer" in nm: continue
    import re; short=re.sub(r"^void ","",re.sub(r"\(anonymous namespace\)::","",nm)).split("(")[0][:58]
    calls=int(r["Calls"]); avg=float(r["AverageNs"])/1e3
    tot+=calls*avg/10
    if calls*avg/10>40: print("%-58s calls/fwd %6.1f avg %7.2f us per-fwd %8.1f us"%(short,calls/10,avg,calls*avg/10))
print("total kernel time per forward: %.1f us"%tot)
PY
