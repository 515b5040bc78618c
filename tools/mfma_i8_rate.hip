// Issue-rate ceiling of the two i8 MFMA shapes on gfx950 (developer tool): independent accumulators, no memory.
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_i8_rate.hip -o /tmp/mfma_i8_rate && /tmp/mfma_i8_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
template <int SHAPE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(int iters, int* out) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {4, 5, (int)blockIdx.x, 7};
    if constexpr (SHAPE == 32) {
        v16i acc[NACC];
        for (int j = 0; j < NACC; ++j)
            for (int r = 0; r < 16; ++r) acc[j][r] = 0;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, acc[j], 0, 0, 0);
        }
        int s = 0;
        for (int j = 0; j < NACC; ++j)
            for (int r = 0; r < 16; ++r) s += acc[j][r];
        if (s == 0x7fffffff) out[0] = s;
    } else {
        v4i acc[NACC];
        for (int j = 0; j < NACC; ++j) acc[j] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[j], 0, 0, 0);
        }
        int s = 0;
        for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
        if (s == 0x7fffffff) out[0] = s;
    }
}
// the big-tile GEMM's inner pattern: 16 accumulators (4 x 4), four A and four B fragments from memory (random bytes), no other work
__global__ __launch_bounds__(256) void tile_kernel(int iters, const v4i* __restrict__ src, int* out) {
    v4i fa[4], fb[4];
    for (int i = 0; i < 4; ++i) fa[i] = src[threadIdx.x + 256 * i], fb[i] = src[threadIdx.x + 256 * (4 + i)];
    v16i acc[4][4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[j][i], 0, 0, 0);
    }
    int s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[j][i][r];
    if (s == 0x7fffffff) out[0] = s;
}
void run_tile(bool random) {
    int* out;
    v4i* src;
    hipMalloc(&out, 4);
    hipMalloc(&src, 256 * 8 * 16);
    unsigned char host[256 * 8 * 16];
    unsigned x = 12345;
    for (auto& c : host) { x = x * 1664525u + 1013904223u; c = random ? (unsigned char)(x >> 24) : 1; }
    hipMemcpy(src, host, sizeof(host), hipMemcpyHostToDevice);
    const int iters = 1000, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    tile_kernel<<<blocks, 256>>>(iters, src, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    tile_kernel<<<blocks, 256>>>(iters, src, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * 16;
    printf("{\"mfma\": \"v_mfma_i32_32x32x32_i8, 4 x 4 accumulators, 4 + 4 fragments\", \"data\": \"%s\", \"ms\": %.3f, \"tops\": %.1f, \"ns_per_mfma_and_simd\": %.2f}\n",
           random ? "random bytes" : "ones", ms, 2 * 32768.0 * n / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * 16));
}
template <int SHAPE, int NACC>
void run(const char* name, int waves_per_cu) {
    int* out;
    hipMalloc(&out, 4);
    const int iters = 4000, blocks = 256 * (waves_per_cu / 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<SHAPE, NACC><<<blocks, 256>>>(iters, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<SHAPE, NACC><<<blocks, 256>>>(iters, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double macs = (SHAPE == 32 ? 32.0 * 32 * 32 : 16.0 * 16 * 64), n = (double)blocks * 4 * iters * NACC;
    printf("{\"mfma\": \"%s\", \"accumulators\": %d, \"waves_per_cu\": %d, \"ms\": %.3f, \"tops\": %.1f, \"ns_per_mfma_and_simd\": %.2f}\n", name, NACC, waves_per_cu, ms,
           2 * macs * n / (ms * 1e-3) / 1e12, ms * 1e6 / (iters * NACC * (waves_per_cu / 4)));
}
int main() {
    run<32, 4>("v_mfma_i32_32x32x32_i8", 4);
    run<32, 8>("v_mfma_i32_32x32x32_i8", 4);
    run<32, 4>("v_mfma_i32_32x32x32_i8", 8);
    run<16, 4>("v_mfma_i32_16x16x64_i8", 4);
    run<16, 8>("v_mfma_i32_16x16x64_i8", 4);
    run<16, 8>("v_mfma_i32_16x16x64_i8", 8);
    run_tile(false);
    run_tile(true);
    return 0;
}
