#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r6f; mkdir -p $OUT; cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $?"
tail -5 $OUT/tests.log
timeout 600 python bench.py --no-yolo > $OUT/bench.json 2> $OUT/bench.log; echo "bench rc $?"
python - <<'P'
import json
d=json.loads(open("gpurun_out/r6f/bench.json").read().strip().splitlines()[-1])
sv=d["sensevoice"]; print({k:sv[k] for k in ("c4_ms_per_step","c3_model_ms","rtf_c4","rtf_model","c4_ms_per_step_exact","plan_statements")})
P
