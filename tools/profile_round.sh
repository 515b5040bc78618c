#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): every measurement a round's profiles/ files are condensed from.
#   usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
#   then here: python tools/summarize_profile.py <tag>
# Counters are collected in their own rocprofv3 passes with --kernel-trace only (no sys / hip / hsa trace domains).
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 0. VALU issue-rate microbenchmark (the ceiling the front-end's pass loop is priced against)
$R/tools/valu_rate > "$OUT/valu_rate.json" 2> "$OUT/valu_rate.err"
# 0b. what the CU-side load path delivers (shared lines, shared lines in different orders, private regions): section 3.3's figure
[ -x $R/tools/l2bw ] && timeout 60 $R/tools/l2bw > "$OUT/l2bw.txt" 2> "$OUT/l2bw.err"
# 1. the default bench invocation: plain, and under rocprofv3 --kernel-trace --stats (front-end leg only under the profiler)
timeout 280 python $R/bench.py > "$OUT/bench_plain.json" 2> "$OUT/bench_plain.log"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python $R/bench.py --no-model --no-cpu-baseline > "$OUT/bench_under_rocprof.json" 2> "$OUT/stats.log"
# 2. HBM traffic of fe_main_kernel: FETCH_SIZE and WRITE_SIZE, one pass each
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$C" -o bench -- python $R/bench.py --no-model --no-cpu-baseline --steps 20 --warmup 5 > "$OUT/pmc_$C.json" 2> "$OUT/pmc_$C.log"
done
# 3. instruction counts of fe_main_kernel (the VALU work the roofline block prices)
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/pmc_sq" -o bench -- \
    python $R/bench.py --no-model --no-cpu-baseline --steps 10 --warmup 2 > "$OUT/pmc_sq.json" 2> "$OUT/pmc_sq.log"
# 4. per-kernel tables of the compiled SenseVoice-shaped plan (configs[2] and one configs[3] shard), 10 eager forwards each
for C in c3 c4; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/sv" -o ${C}_compiled -- \
      python $R/tools/sensevoice_graph.py --compiled-only --configs $C --runs 8 > "$OUT/sv_${C}.json" 2> "$OUT/sv_${C}.log"
done
# 4b. configs[4]: the Yolo26n-seg-shaped network at batch 64 as one compiled graph (per-image check against the batch-1 plan inside)
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/yolo" -o n64 -- \
    python $R/tools/yolo_graph.py --batch 64 --check 4 --runs 5 --out "$OUT/yolo_n64_under_rocprof.json" > "$OUT/yolo_prof.log" 2>&1
timeout 300 python $R/tools/yolo_graph.py --batch 64 --check 8 --table "$OUT/yolo_table.json" --out "$OUT/yolo_n64.json" > "$OUT/yolo.log" 2>&1
# 4c. the reference's own generated Yolo26n-seg graph, re-batched to 64 images as one graph (where the lifted plan exists)
[ -f $R/_lifted/yolo26seg_plan.json ] && timeout 300 python $R/tools/yolo_lifted_batch.py --batch 64 --table "$OUT/yolo_lifted_table.json" --out "$OUT/yolo_lifted_n64.json" > "$OUT/yolo_lifted.log" 2>&1
# 4d. ConvInteger: the i8 route against lele's f32 formulation (lab build), and the exhaustive check of the short reciprocal
if [ -f $R/lele_amd/liblele_hip_lab.so ]; then
  LELE_HIP_LAB=1 timeout 200 python $R/tools/conv_integer_bench.py > "$OUT/conv_integer_i8.json" 2> "$OUT/conv_integer.log"
  LELE_HIP_LAB=1 LELE_HIP_CONV_INTEGER_F32=1 timeout 200 python $R/tools/conv_integer_bench.py > "$OUT/conv_integer_f32.json" 2>> "$OUT/conv_integer.log"
fi
[ -x $R/tools/recip_check ] && timeout 120 $R/tools/recip_check > "$OUT/recip_check.json" 2> "$OUT/recip_check.err"
# 5. operator micro-benchmarks
timeout 280 python $R/tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.log" 2>&1
timeout 200 python $R/tools/qlinear_bench.py --out "$OUT/qlinear.json" > "$OUT/qlinear.log" 2>&1
# 6. the fused attention kernel and the persistent i8 GEMM from the inside (cycle-counter stamps per phase), and their timings
timeout 120 python $R/tools/attention_bench.py > "$OUT/attention_bench.json" 2> "$OUT/attention_bench.log"
if [ -f $R/lele_amd/liblele_hip_lab.so ]; then
  LELE_HIP_LAB=1 timeout 120 python $R/tools/attention_stamps.py > "$OUT/attention_stamps.txt" 2> "$OUT/attention_stamps.log"
fi
# 6b. the attention kernels' matrix-core and vector-pipe occupancy in ONE counter pass (kernel-trace only beside it): does the
#     MFMA of one tile issue under the VALU of another?
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU --kernel-trace --output-format csv \
    -d "$OUT/pmc_attn" -o attn -- python $R/tools/attention_bench.py --only default --reps 40 > "$OUT/pmc_attn.json" 2> "$OUT/pmc_attn.log"
timeout 120 python $R/tools/rs_bench.py > "$OUT/rs_bench.txt" 2> "$OUT/rs_bench.log"
if [ -f $R/lele_amd/liblele_hip_lab.so ]; then  # stamps exist in the lab build only
  LELE_HIP_LAB=1 timeout 120 python $R/tools/rs_stamps.py > "$OUT/rs_stamps.txt" 2> "$OUT/rs_stamps.log"
fi
# 7. the sharded recogniser step through the C ABI alone: the native runner on a 2-layer SenseVoice-shaped plan, one rank per visible
#    GPU (fork before any HIP call, file rendezvous, RCCL all-gather of the decoded ids through lele_hip_comm_*)
NGPU=$(python3 -c "import torch; print(max(1, torch.cuda.device_count()))" 2>/dev/null || echo 1)
( cd $R && timeout 250 python3 - "$OUT" <<'PY'
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import lele_amd
from lele_amd.compiler import compile_model
from sensevoice_graph import Encoder, encoder_onnx
out = sys.argv[1]
ctx = lele_amd._lib.Ctx(0)
enc = Encoder(ctx, 2)
plan, blob = compile_model(encoder_onnx(enc, 8), "sensevoice_shaped")
json.dump(plan, open(os.path.join(out, "sv2_plan.json"), "w"))
open(os.path.join(out, "sv2_weights.bin"), "wb").write(blob)
np.random.default_rng(0).standard_normal((8, 167, 560)).astype(np.float32).tofile(os.path.join(out, "sv2_feats.bin"))
PY
) > "$OUT/lele_run_prepare.log" 2>&1
timeout 200 $R/lele_amd/lele_run "$OUT/sv2_plan.json" "$OUT/sv2_weights.bin" --input feats="$OUT/sv2_feats.bin":f32:8,167,560 --out "$OUT/sv2_out" \
    --runs 5 --ranks $NGPU --decode > "$OUT/lele_run_ranks.json" 2> "$OUT/lele_run_ranks.log"
rm -f "$OUT/sv2_weights.bin" "$OUT"/sv2_out*.bin "$OUT/sv2_feats.bin"
find "$OUT" -name '*.csv' | wc -l
tail -c 400 "$OUT/bench_plain.json"
