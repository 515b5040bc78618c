// tools/mfma_valu_overlap.hip -- does the matrix pipe run BESIDE the vector pipe on gfx950?  (the staged matrix-core
// DFT of the front-end is only worth costing if its products hide behind the vector work that remains.)
// Every SIMD gets `w` waves; a wave runs ITER rounds of M MFMAs (v_mfma_f32_32x32x16_bf16, 4 independent accumulators) and / or V plain
// v_fma_f32 (16 independent chains).  Cases: MFMA only, VALU only, both in the SAME wave (interleaved), and MFMA waves beside VALU waves
// (even waves multiply, odd waves do vector work).  Prints ns per round and SIMD; perfect overlap = max of the two "only" figures.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/mfma_valu_overlap && ./tools/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int M = 8, V = 64;   // per round: 8 MFMAs (8 x 32 cycles of the matrix pipe at peak clock), 64 v_fma (64 x 4 cycles of the vector pipe)
template <int MODE>  // 0 MFMA only, 1 VALU only, 2 both in one wave, 3 even waves MFMA / odd waves VALU
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    // (MODE 3: a workgroup is four waves, one a SIMD; workgroups alternate in groups of 256 (one a CU: the dispatcher deals them round robin over XCDs and CUs) between multiplying and vector work: every SIMD hosts both kinds)
    const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && ((blockIdx.x >> 8) & 1) == 0);
    const bool do_v = MODE == 1 || MODE == 2 || (MODE == 3 && ((blockIdx.x >> 8) & 1) == 1);
    f32x16 acc[4];
    float a[16];
    bf16x8 x, y;
    for (int i = 0; i < 8; ++i) x[i] = (__bf16)(seed + i + (threadIdx.x & 7)), y[i] = (__bf16)(seed * 0.5f + i);
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) acc[j][r] = seed;
    for (int i = 0; i < 16; ++i) a[i] = seed + i + threadIdx.x;
    const float b = seed + 1.0f, c = seed * 0.25f;
    if (MODE == 3) {   // two separate loops: no branch inside a round
        if (do_m) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
            }
        } else {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[v & 15]) : "v"(b), "v"(c));
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
        if (do_m && do_v) {
#pragma unroll
            for (int m = 0; m < M; ++m) {
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
#pragma unroll
                for (int v = 0; v < V / M; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[(m * (V / M) + v) & 15]) : "v"(b), "v"(c));
            }
        } else if (do_m) {
#pragma unroll
            for (int m = 0; m < M; ++m) acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[m & 3], 0, 0, 0);
        } else if (do_v) {
#pragma unroll
            for (int v = 0; v < V; ++v) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[v & 15]) : "v"(b), "v"(c));
        }
    }
    float s = 0.0f;
    for (int j = 0; j < 4; ++j)
        for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 16; ++i) s += a[i];
    if (s == 123.456f) out[0] = s;
}
template <int MODE>
static float run(float* d, int cus, int wps) {
    const int iters = 20000, blocks = cus * wps;   // 256 threads = 4 waves: one a SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best * 1e6f / iters;   // ns per round (all waves of a SIMD together)
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* d;
    hipMalloc(&d, 4);
    const int cus = p.multiProcessorCount;
    printf("{\"what\": \"ns per round and SIMD; a round = %d v_mfma_f32_32x32x16_bf16 and / or %d v_fma_f32 per wave\", \"cus\": %d,\n", M, V, cus);
    for (int wps = 2; wps <= 4; wps += 2) {
        const float m = run<0>(d, cus, wps), v = run<1>(d, cus, wps), both = run<2>(d, cus, wps), side = run<3>(d, cus, wps);
        printf(" \"waves_per_simd_%d\": {\"mfma_only\": %.1f, \"valu_only\": %.1f, \"both_in_one_wave\": %.1f, \"mfma_waves_beside_valu_waves\": %.1f,"
               " \"note\": \"side by side half the waves do each kind: compare with half of each only-figure\"}%s\n",
               wps, m, v, both, side, wps == 4 ? "" : ",");
    }
    printf("}\n");
    return 0;
}
