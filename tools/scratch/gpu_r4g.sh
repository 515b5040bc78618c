#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4g
O=gpurun_out/r4g
( timeout 600 python -m pytest tests/test_lift_generated.py -m gpu -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -E "^(FAILED|ERROR)|passed|failed|rc=|Error" $O/tests.log | tail -6
( timeout 600 python tools/yolo_lifted_batch.py --batch 64 --check 2 --table $O/lifted_table.json --out $O/yolo26seg_lifted_n64.json > $O/lifted.log 2> $O/lifted_table.txt; echo "rc=$?" >> $O/lifted.log )
tail -2 $O/lifted.log; cat $O/lifted_table.txt | head -30
python - <<PY
import json
d=json.load(open("$O/lifted_table.json"))
for r in d["slowest_statements"][:30]: print(r)
print(d["total_ms"])
PY
