// tools/valu_rate.hip -- VALU issue-rate microbenchmark for gfx950 (the ceiling the front-end's pass loop is priced against).
//
// One workgroup of `waves` waves per SIMD slot (grid = CUs x 4 x waves / 4 workgroups of 256 threads, or one wave per
// workgroup when waves < 4 ...): every wave runs ITER iterations of 32 independent chains of one instruction (inline asm, so
// the compiler can neither fuse nor drop them).  Prints cycles per wave-instruction per SIMD at the measured clock
// (wall_clock vs s_memtime) for plain and packed f32 ops, and the latency of a dependent chain.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rate.hip -o tools/valu_rate && ./tools/valu_rate > profiles/r02_valu_rate.json
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(X) X X X X X X X X
#define CHAINS 16

template <int OP>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed, unsigned long long* cyc) {
    float a[CHAINS];
    f2 p[CHAINS];
    const float b = seed + 1.0f, c = seed * 0.5f;
    const f2 pb = {b, b}, pc = {c, c};
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) {
        a[i] = seed + i + threadIdx.x;
        p[i] = (f2){a[i], a[i] + 1.0f};
    }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < CHAINS; ++i) {
            if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
            if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
            if (OP == 4) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
            if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));  // one dependent chain: latency
            if (OP == 7) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(b));
            if (OP == 8) asm volatile("v_fma_f32 %0, %0, %1, %2\n v_add_f32 %3, %3, %1" : "+v"(a[i]), "+v"(p[i].x) : "v"(b), "v"(c));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += a[i] + p[i].x + p[i].y;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int OP>
void run(const char* name, int instr_per_iter, float* d, unsigned long long* dc, int cus, bool last) {
    const int iters = 20000;
    printf("  \"%s\": {", name);
    for (int wps = 1; wps <= 4; wps *= 2) {  // waves per SIMD
        const int blocks = cus * wps;      // 256 threads = 4 waves = one per SIMD of a CU (the dispatcher spreads them)
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {  // best of five back-to-back launches: the first ones see ramping clocks
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0f, dc);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double instr_per_simd = (double)iters * instr_per_iter * wps;
        printf("\"wps%d\": {\"ms\": %.4f, \"ns_per_instr_per_simd\": %.4f}%s", wps, best, best * 1e6 / instr_per_simd, wps < 4 ? ", " : "");
    }
    printf("}%s\n", last ? "" : ",");
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    float* d;
    unsigned long long* dc;
    hipMalloc(&d, 64);
    hipMalloc(&dc, 64);
    printf("{\n  \"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d,\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    printf("  \"note\": \"ns_per_instr_per_simd = best-of-5 wall time / wave-instructions issued per SIMD (wps = waves per SIMD); a warm-up launch of ~50 ms precedes the table\",\n");
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(rate_kernel<0>, dim3(prop.multiProcessorCount * 4), dim3(256), 0, 0, d, 20000, 1.0f, dc);
    hipDeviceSynchronize();
    const int cus = prop.multiProcessorCount;
    run<0>("v_fma_f32", CHAINS, d, dc, cus, false);
    run<1>("v_add_f32", CHAINS, d, dc, cus, false);
    run<2>("v_mul_f32", CHAINS, d, dc, cus, false);
    run<3>("v_pk_fma_f32", CHAINS, d, dc, cus, false);
    run<4>("v_pk_add_f32", CHAINS, d, dc, cus, false);
    run<5>("v_pk_mul_f32", CHAINS, d, dc, cus, false);
    run<6>("v_add_f32_dependent_chain", CHAINS, d, dc, cus, false);
    run<7>("v_mov_b32", CHAINS, d, dc, cus, false);
    run<8>("v_fma_f32+v_add_f32_pair", 2 * CHAINS, d, dc, cus, true);
    printf("}\n");
    return 0;
}
