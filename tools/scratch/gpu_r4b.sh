#!/bin/bash
# round 4, GPU call B (runs ON the GPU box): the whole GPU suite, the bench line, the Yolo-shaped per-layer table, the C4 kernel table
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4b
O=gpurun_out/r4b
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
tail -30 $O/tests.log
( timeout 600 python tools/yolo_graph.py --batch 64 --check 2 --table $O/yolo_table.json --out $O/yolo_n64.json > $O/yolo.log 2> $O/yolo_table.txt; echo "rc=$?" >> $O/yolo.log )
tail -3 $O/yolo.log; head -60 $O/yolo_table.txt
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
tail -c 2500 $O/bench.json; tail -3 $O/bench.err
bash tools/kstats_sv.sh c4 r4b_c4 > $O/kstats_c4.txt 2>&1
head -30 $O/kstats_c4.txt
