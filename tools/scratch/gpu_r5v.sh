#!/bin/bash
# front-end: sample prefetch right after the samples are consumed (p0) against after phase A (p3): time and fabric traffic
for l in ab_fe_p3.so liblele_hip.so; do echo -n "$l "; PYTHONPATH=. LELE_HIP_LIBRARY=$l python tools/scratch/fe_hash.py 2>&1 | tail -1; done
run() { echo -n "$1 "; LELE_HIP_LIBRARY=$1 timeout 200 python bench.py --no-model --no-yolo --no-cpu-baseline --steps 100 --warmup 10 2>&1 | tail -1 | grep -o '"value": [0-9.]*, \|"kernel_ms": [0-9.]*' | tr '\n' ' '; echo; }
for i in 1 2; do run ab_fe_p3.so; run liblele_hip.so; done
cd /tmp; export TMPDIR=/tmp
for l in ab_fe_p3.so liblele_hip.so; do
for C in FETCH_SIZE WRITE_SIZE; do
LELE_HIP_LIBRARY=$l timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r5v_${l}_$C -o k -- python $GRAFT_REPO_ROOT/bench.py --no-model --no-yolo --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
python - <<PY
import csv, collections
v = collections.defaultdict(list)
for r in csv.DictReader(open("$GRAFT_REPO_ROOT/gpurun_out/r5v_${l}_$C/k_counter_collection.csv")):
    if "fe_main" in r["Kernel_Name"]: v[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("$l", {k: round(sum(x)/len(x)) for k, x in v.items()})
PY
done
done
