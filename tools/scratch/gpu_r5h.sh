#!/bin/bash
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_real_audio.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do timeout 300 python bench.py --no-model --no-yolo --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c "import json,sys; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['roofline'].get('kernel_ms'))"; done
