#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): every measurement a round's profiles/ files are condensed from.
#   usage: tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/...
#   then here: python tools/summarize_profile.py <tag>
# Counters are collected in their own rocprofv3 passes with --kernel-trace only (no sys / hip / hsa trace domains).
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# 0. VALU issue-rate microbenchmark (the ceiling the front-end's pass loop is priced against)
$R/tools/valu_rate > "$OUT/valu_rate.json" 2> "$OUT/valu_rate.err"
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
# 5. operator micro-benchmarks
timeout 280 python $R/tools/microbench.py --out "$OUT/microbench.json" > "$OUT/microbench.log" 2>&1
timeout 200 python $R/tools/qlinear_bench.py --out "$OUT/qlinear.json" > "$OUT/qlinear.log" 2>&1
# 6. the fused attention kernel and the persistent i8 GEMM from the inside (cycle-counter stamps per phase), and their timings
timeout 120 python $R/tools/attention_bench.py > "$OUT/attention_bench.json" 2> "$OUT/attention_bench.log"
timeout 120 python $R/tools/attention_stamps.py > "$OUT/attention_stamps.txt" 2> "$OUT/attention_stamps.log"
timeout 120 python $R/tools/wholek_stamps.py > "$OUT/wholek_stamps.txt" 2> "$OUT/wholek_stamps.log"
# 7. the sharded recogniser step through the C ABI alone (native runner, RCCL group of one rank per visible GPU)
timeout 200 $R/lele_amd/lele_run --help > /dev/null 2>&1
find "$OUT" -name '*.csv' | wc -l
tail -c 400 "$OUT/bench_plain.json"
