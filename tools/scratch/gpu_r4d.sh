#!/bin/bash
# round 4, GPU call D: the LayerNorm group kernel (tests, C4 table), Yolo-shaped graph with the stride-2 32-channel blocks and 1 x 1 thresholds
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4d
O=gpurun_out/r4d
( timeout 900 python -m pytest tests/test_quant.py tests/test_fullsize_graph.py tests/test_compiler.py tests/test_conv_rnn.py tests/test_fullsize_properties.py tests/test_channel_views.py -m gpu --maxfail=8 -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/tests.log | tail -12
bash tools/kstats_sv.sh c4 r4d_c4 > $O/kstats_c4.txt 2>&1
head -16 $O/kstats_c4.txt
for cfg in "64 6400 48" "128 6400 48" "128 1600 32" "256 1600 32" "256 400 32"; do
  set -- $cfg
  ( LELE_HIP_CONV_W1_MAXOC=$1 LELE_HIP_CONV_W1_MINPLANE=$2 LELE_HIP_CONV_W1_MINC=$3 timeout 300 python tools/yolo_graph.py --batch 64 --no-batch1 --runs 10 --table $O/yolo_table_$1_$2.json > $O/yolo_$1_$2.log 2> $O/yolo_table_$1_$2.txt )
  python - <<PY
import json
d = json.loads(open("$O/yolo_$1_$2.log").read().strip().splitlines()[-1])
print("W1 maxoc $1 minplane $2 minc $3: graph_ms", d["graph_ms_per_forward"])
PY
done
grep " k1 " $O/yolo_table_64_6400.txt | head -24
echo ----
grep " k1 " $O/yolo_table_256_400.txt | head -24
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "c4_ms", d["sensevoice"]["c4_ms_per_step"], "c3_ms", d["sensevoice"]["c3_model_ms"], "yolo_ms", d["yolo"]["ms_per_forward"])
PY
tail -2 $O/bench.err
