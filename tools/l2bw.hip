// l2bw.hip -- what the CU-side load path delivers when many CUs read the SAME few hundred KB (weights / a shared activation panel)
// hipcc -O3 --offload-arch=gfx950 tools/l2bw.hip -o tools/l2bw   (the binary is git-ignored; it travels to the GPU box with the tree)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
// every wave sweeps `bytes` of the buffer `sweeps` times, 16 x 1 KiB loads in flight; MODE 0: all waves the same addresses in the
// same order; MODE 1: each wave starts at its own offset (same set of lines, different order); MODE 2: each workgroup its own
// private region (bytes per workgroup; distinct lines)
template <int MODE>
__global__ __launch_bounds__(512) void sweep(const v4i* buf, size_t bytes, int sweeps, int* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t nblk = bytes / 1024;  // 1 KiB blocks
    const v4i* base = buf + (MODE == 2 ? (size_t)blockIdx.x * (bytes / 16) : 0);
    size_t start = MODE == 1 ? ((size_t)(blockIdx.x * 8 + wave) * 37) % nblk : 0;
    v4i acc = {0, 0, 0, 0};
    for (int s = 0; s < sweeps; ++s)
        for (size_t b = 0; b < nblk; b += 16) {
            v4i v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = base[((start + b + i) % nblk) * 64 + lane];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc += v[i];
        }
    if (acc[0] + acc[1] + acc[2] + acc[3] == 0x12345678) sink[0] = 1;
}
int main() {
    const size_t cap = 512u << 20;
    v4i* d;
    int* sink;
    hipMalloc(&d, cap);
    hipMalloc(&sink, 4);
    hipMemset(d, 1, cap);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto run = [&](int mode, size_t bytes, int sweeps, int grid) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(sweep<0>, dim3(grid), dim3(512), 0, 0, d, bytes, sweeps, sink);
            if (mode == 1) hipLaunchKernelGGL(sweep<1>, dim3(grid), dim3(512), 0, 0, d, bytes, sweeps, sink);
            if (mode == 2) hipLaunchKernelGGL(sweep<2>, dim3(grid), dim3(512), 0, 0, d, bytes, sweeps, sink);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double tot = (double)bytes * sweeps * grid * 8;
        printf("{\"mode\": %d, \"bytes\": %zu, \"sweeps\": %d, \"grid\": %d, \"us\": %.1f, \"cu_side_TBps\": %.2f, \"B_per_clk_per_CU_at_2.1GHz\": %.1f}\n", mode, bytes,
               sweeps, grid, ms * 1e3, tot / ms / 1e9, tot / (ms * 1e-3) / 256 / 2.1e9);
    };
    for (int mode = 0; mode < 2; ++mode)
        for (size_t kb : {128, 768, 4096, 32768}) run(mode, kb << 10, (int)(16384 / kb > 0 ? 16384 / kb : 1) * 2, 256);
    for (size_t kb : {16, 128, 1024}) run(2, kb << 10, (int)(8192 / kb), 256);
    // one sweep only (cold start, the prologue-burst case): 128 KiB per wave
    run(0, 128 << 10, 1, 256);
    run(1, 128 << 10, 1, 256);
    run(1, 768 << 10, 1, 256);
    return 0;
}
