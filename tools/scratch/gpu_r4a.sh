#!/bin/bash
# round 4, GPU call A (runs ON the GPU box): the new tests, the bench line, the Yolo-shaped per-layer table and kernel trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4a
O=gpurun_out/r4a
python -c "import torch" 2>/dev/null
( timeout 900 python -m pytest tests/test_channel_views.py tests/test_manip.py tests/test_conv_rnn.py tests/test_conv_integer.py tests/test_fullsize_properties.py tests/test_fuzz_gpu.py -m gpu -x -q > $O/tests_new.log 2>&1; echo "rc=$?" >> $O/tests_new.log ) 
tail -5 $O/tests_new.log
( timeout 600 python tools/yolo_graph.py --batch 64 --check 2 --table $O/yolo_table.json --out $O/yolo_n64.json > $O/yolo.log 2> $O/yolo_table.txt; echo "rc=$?" >> $O/yolo.log )
tail -3 $O/yolo.log; head -45 $O/yolo_table.txt
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
tail -c 3000 $O/bench.json; tail -3 $O/bench.err
bash tools/kstats.sh r4a_yolo python tools/yolo_graph.py --batch 64 --no-batch1 --runs 10 > $O/kstats.txt 2>&1
head -40 $O/kstats.txt
