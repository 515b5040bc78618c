#!/bin/bash
# round 4, call H: radix-select top-k and the x4 / x8 up-sampling, then the two forwards and the bench line
mkdir -p gpurun_out/r4h
timeout 600 python -m pytest tests/test_manip.py tests/test_lift_generated.py tests/test_channel_views.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r4h/tests.log
echo rc=$?
timeout 400 python tools/yolo_lifted_batch.py --batch 64 --table gpurun_out/r4h/lifted_table.json --out gpurun_out/r4h/yolo26seg_lifted_n64.json 2>&1 | tail -30 | tee gpurun_out/r4h/lifted.log
timeout 400 python tools/yolo_graph.py --batch 64 --no-batch1 --table gpurun_out/r4h/yolo_table.json --out gpurun_out/r4h/yolo_n64.json 2>&1 | tail -24 | tee gpurun_out/r4h/yolo.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r4h/bench.json 2> gpurun_out/r4h/bench.err
tail -c 3000 gpurun_out/r4h/bench.json
