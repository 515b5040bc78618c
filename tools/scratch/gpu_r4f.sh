#!/bin/bash
# round 4, GPU call F: the reference's generated Yolo26n-seg graph at batch 64, depthwise direct stores, bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4f
O=gpurun_out/r4f
( timeout 1200 python -m pytest tests/test_lift_generated.py tests/test_conv_rnn.py tests/test_fullsize_properties.py tests/test_channel_views.py tests/test_compiler.py -m gpu --maxfail=8 -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error" $O/tests.log | tail -12
( timeout 600 python tools/yolo_lifted_batch.py --batch 64 --check 4 --out $O/yolo26seg_lifted_n64.json > $O/lifted.log 2>&1; echo "rc=$?" >> $O/lifted.log )
tail -3 $O/lifted.log
( timeout 600 python tools/yolo_graph.py --batch 64 --check 2 --table $O/yolo_table.json --out $O/yolo_n64.json > $O/yolo.log 2> $O/yolo_table.txt; echo "rc=$?" >> $O/yolo.log )
python - <<PY
import json
d = json.loads(open("$O/yolo.log").read().strip().splitlines()[0])
print("yolo graph_ms", d["graph_ms_per_forward"], d["max_error_in_units_of_1e-4_per_output"], d["folded_equals_unfolded_bitwise"])
PY
grep "g80\|g64\|g128\|g256\|@20x20" $O/yolo_table.txt | head -12
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "c4_ms", d["sensevoice"]["c4_ms_per_step"], "c3_ms", d["sensevoice"]["c3_model_ms"], "yolo_ms", d["yolo"]["ms_per_forward"])
print(json.dumps(d["yolo"].get("reference_graph"))[:900])
PY
tail -2 $O/bench.err
