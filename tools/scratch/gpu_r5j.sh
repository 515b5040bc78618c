#!/bin/bash
timeout 900 python -m pytest tests/test_frontend_gpu.py tests/test_real_audio.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 400 python tools/microbench.py --only frontend --out gpurun_out/mb_frontend.json 2>&1 | tail -3 | cut -c1-250
