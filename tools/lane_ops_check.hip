// Device check of lele_amd/csrc/lane_ops.h against the __shfl forms it replaces (wave-relative thread ids: run with 64-thread
// and 256-thread blocks).  Build: hipcc --offload-arch=gfx950 -I lele_amd/csrc tools/lane_ops_check.hip -o tools/lane_ops_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lane_ops.h"
using namespace lele;

__global__ void check(const float* in, int* bad) {
    const int t = threadIdx.x, l = t & 31;
    const float x = in[blockIdx.x * blockDim.x + t];
    int b = 0;
    auto same = [&](float a, float c) { return __float_as_uint(a) == __float_as_uint(c); };
    if ((l & 15) + 8 < 16 && !same(row_down<8>(x), __shfl_down(x, 8, 32))) b |= 1;
    if ((l & 15) + 4 < 16 && !same(row_down<4>(x), __shfl_down(x, 4, 32))) b |= 2;
    if ((l & 15) + 2 < 16 && !same(row_down<2>(x), __shfl_down(x, 2, 32))) b |= 4;
    if ((l & 15) + 1 < 16 && !same(row_down<1>(x), __shfl_down(x, 1, 32))) b |= 8;
    if (!same(swap16(x), __shfl_xor(x, 16, 32))) b |= 16;
    float m = x;
    for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 32));
    if (!same(group_max32(x), m)) b |= 32;
    for (int idx = 0; idx < 32; ++idx)
        if (!same(group_read(x, idx), __shfl(x, idx, 32))) b |= 64;
    if (!same(swap32(x), __shfl_xor(x, 32))) b |= 128;
    float mw = x;
    for (int off = 32; off > 0; off >>= 1) mw = fmaxf(mw, __shfl_xor(mw, off));
    if (!same(wave_allreduce64(x, [](float a, float c) { return fmaxf(a, c); }), mw)) b |= 256;
    int si = (int)(x * 3.0f);
    int sw = si;
    for (int off = 32; off > 0; off >>= 1) sw += __shfl_xor(sw, off);
    if (wave_sum_i32(si) != sw) b |= 512;
    if (b) atomicOr(bad, b);
}

int main() {
    const int n = 256 * 64;
    float* h = new float[n];
    unsigned s = 12345;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u;
        h[i] = (float)((int)(s >> 8) - (1 << 23)) / 1024.0f;
    }
    float* d;
    int* bad;
    hipMalloc(&d, n * 4);
    hipMalloc(&bad, 4);
    hipMemcpy(d, h, n * 4, hipMemcpyHostToDevice);
    int total = 0;
    for (int threads : {64, 256}) {
        hipMemset(bad, 0, 4);
        hipLaunchKernelGGL(check, dim3(n / threads), dim3(threads), 0, 0, d, bad);
        int b = -1;
        hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
        printf("blocks of %d threads: mismatch mask %d\n", threads, b);
        total |= b;
    }
    printf(total ? "FAILED\n" : "lane_ops: all forms agree with __shfl\n");
    return total != 0;
}
