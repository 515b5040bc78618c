#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/r6c; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_quant.py tests/test_fullsize_graph.py tests/test_compiler.py tests/test_native_runner.py tests/test_lanes.py tests/test_comm_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1; echo "tests rc $?"
tail -15 $OUT/tests.log
bash tools/kstats_sv.sh c4 r6c > $OUT/kstats_c4.txt 2>&1; cat $OUT/kstats_c4.txt | tail -14
grep -o '"c4[^}]*' $OUT/sv_c4.json | head -5
