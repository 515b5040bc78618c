#!/bin/bash
mkdir -p gpurun_out/r4v
timeout 900 python -m pytest tests/test_conv_integer.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r4v/tests.log
LELE_HIP_LAB=1 timeout 300 python tools/conv_integer_bench.py | tee gpurun_out/r4v/conv_integer_i8.json
LELE_HIP_LAB=1 LELE_HIP_CONV_INTEGER_F32=1 timeout 300 python tools/conv_integer_bench.py | tee gpurun_out/r4v/conv_integer_f32.json
cd /tmp && export TMPDIR=/tmp && LELE_HIP_LAB=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4v/prof -o ci -- python $GRAFT_REPO_ROOT/tools/conv_integer_bench.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r4v/prof/**/*kernel_stats.csv',recursive=True)
for r in list(csv.DictReader(open(f[0])))[:12]: print(r['Name'][:70], r['Calls'], r['AverageNs'], r['Percentage'])
PY
