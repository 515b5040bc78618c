#!/bin/bash
# round 6, call 1: state of the tree on this round's box (tests, bench line, configs[3] kernel table, i8 compute-bound sweep)
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r6a; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $?" 
tail -3 $OUT/tests.log
timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.log; echo "bench rc $?"
bash tools/kstats_sv.sh c4 r6a > $OUT/kstats_c4.txt 2>&1
bash tools/kstats_sv.sh c3 r6a > $OUT/kstats_c3.txt 2>&1
timeout 500 python tools/qlinear_bench.py --compute-bound --calls 4 --out $OUT/qlinear_compute_bound.json > $OUT/qcb.log 2>&1; echo "qcb rc $?"
tail -5 $OUT/qcb.log
cat $OUT/kstats_c4.txt
