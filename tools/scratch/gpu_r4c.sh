#!/bin/bash
# round 4, GPU call C (runs ON the GPU box): whole GPU suite, front-end A/B (four waves per SIMD), bench line, tables
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
( timeout 1800 python -m pytest tests -m gpu --maxfail=12 -q > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log )
grep -E "^(FAILED|ERROR)|passed|failed|rc=" $O/tests.log | tail -20
( LELE_HIP_FE_W4=1 timeout 600 python -m pytest tests/test_frontend_gpu.py tests/test_real_audio.py -m gpu -q > $O/tests_w4.log 2>&1; echo "rc=$?" >> $O/tests_w4.log )
tail -4 $O/tests_w4.log
for w in 0 1; do
  ( LELE_HIP_FE_W4=$w timeout 300 python bench.py --no-model --no-yolo --no-cpu-baseline --steps 40 --warmup 10 > $O/bench_fe_w4_$w.json 2> $O/bench_fe_w4_$w.err )
  python - <<PY
import json
try:
    d = json.loads(open("$O/bench_fe_w4_$w.json").read().strip().splitlines()[-1])
    print("W4=$w value %.1f GB/s ms_per_step %.4f kernel_ms %.4f" % (d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"]))
except Exception as e:
    print("W4=$w failed", e)
PY
done
( timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err )
python - <<PY
import json
d = json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "c4_ms", d["sensevoice"]["c4_ms_per_step"], "c3_ms", d["sensevoice"]["c3_model_ms"], "yolo_ms", d["yolo"]["ms_per_forward"], d["yolo"].get("channel_views"))
PY
tail -2 $O/bench.err
bash tools/kstats_sv.sh c4 r4c_c4 > $O/kstats_c4.txt 2>&1
head -24 $O/kstats_c4.txt
( timeout 600 python tools/yolo_graph.py --batch 64 --check 2 --table $O/yolo_table.json --out $O/yolo_n64.json > $O/yolo.log 2> $O/yolo_table.txt; echo "rc=$?" >> $O/yolo.log )
tail -1 $O/yolo.log; head -30 $O/yolo_table.txt; tail -1 $O/yolo_table.txt
