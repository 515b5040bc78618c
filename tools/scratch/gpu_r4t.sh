#!/bin/bash
# round 4, call T: the whole GPU suite and the default bench line
mkdir -p gpurun_out/r4t
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r4t/tests.log
timeout 900 python bench.py > gpurun_out/r4t/bench.json 2> gpurun_out/r4t/bench.err
tail -c 1500 gpurun_out/r4t/bench.json
python - <<'PY'
import json
r=json.loads(open('gpurun_out/r4t/bench.json').read().strip().splitlines()[-1])
print({k:r.get(k) for k in ('value','ms_per_step','yolo_ms_per_forward')}, r.get('roofline',{}).get('frac'))
y=r.get('yolo') or {}
print({k:y.get(k) for k in ('ms_per_forward','roofline','channel_views')})
print((y.get('reference_graph') or {}).get('ms_per_forward'))
PY
