# lab build: ablations of igemm_big_kernel (LELE_HIP_IGEMM_FLAGS: 16 no fetch, 32 no MFMA, 64 no epilogue, 128 no LDS stores, 256 no barrier)
for f in ${@:-0 16 32 48 64}; do echo -n "FLAGS=$f "; LELE_HIP_LAB=1 LELE_HIP_IGEMM_FLAGS=$f bash tools/kstats.sh big$f python tools/qlinear_bench.py --compute-bound --shapes 8192x4096x4096 --calls 10 2>&1 | grep igemm_big | cut -c100-170; done
