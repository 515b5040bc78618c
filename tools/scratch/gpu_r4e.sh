#!/bin/bash
# round 4, GPU call E: whole suite on the cleaned tree, bench, Yolo-shaped table and kernel trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4e
O=gpurun_out/r4e
( timeout 1800 python -m pytest tests -m gpu --maxfail=12 -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/tests.log | tail -12
( timeout 600 python tools/yolo_graph.py --batch 64 --check 2 --table $O/yolo_table.json --out $O/yolo_n64.json > $O/yolo.log 2> $O/yolo_table.txt; echo "rc=$?" >> $O/yolo.log )
python - <<PY
import json
d = json.loads(open("$O/yolo.log").read().strip().splitlines()[0])
print("yolo graph_ms", d["graph_ms_per_forward"], d["max_error_in_units_of_1e-4_per_output"], d["folded_equals_unfolded_bitwise"])
PY
head -36 $O/yolo_table.txt; tail -1 $O/yolo_table.txt
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "c4_ms", d["sensevoice"]["c4_ms_per_step"], "c3_ms", d["sensevoice"]["c3_model_ms"], "yolo_ms", d["yolo"]["ms_per_forward"], d["cpu_baseline_all_cores"])
PY
tail -2 $O/bench.err
bash tools/kstats.sh r4e_yolo python tools/yolo_graph.py --batch 64 --no-batch1 --no-fold --runs 10 > /dev/null 2>&1
bash tools/kstats.sh r4e_yolo python tools/yolo_graph.py --batch 64 --no-batch1 --runs 10 > $O/kstats.txt 2>&1
head -32 $O/kstats.txt
