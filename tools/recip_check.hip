// Exhaustive check of the short reciprocal the SiLU / sigmoid epilogues use (lele_amd/csrc/simd_math.h, recip_ge1) against the IEEE
// division lele's AVX2 code performs (_mm256_div_ps: correctly rounded): every f32 d in [1, 2^126], and the whole SiLU for EVERY f32 v.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -Ilele_amd/csrc tools/recip_check.hip -o tools/recip_check && tools/recip_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include "simd_math.h"
using namespace lele;

__device__ __forceinline__ float recip_a(float d) {  // one Newton step
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf_(fmaf_(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float recip_b(float d) {  // two
    float r = __builtin_amdgcn_rcpf(d);
    r = fmaf_(fmaf_(-d, r, 1.0f), r, r);
    return fmaf_(fmaf_(-d, r, 1.0f), r, r);
}
__global__ void check_recip(unsigned lo, unsigned hi, unsigned long long* bad) {
    unsigned long long a = 0, b = 0, raw = 0;
    for (unsigned long long u = lo + blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; u <= hi; u += (unsigned long long)gridDim.x * blockDim.x) {
        const float d = __uint_as_float((unsigned)u), want = 1.0f / d;
        a += __float_as_uint(recip_a(d)) != __float_as_uint(want);
        b += __float_as_uint(recip_b(d)) != __float_as_uint(want);
        raw += __float_as_uint(__builtin_amdgcn_rcpf(d)) != __float_as_uint(want);
    }
    if (a) atomicAdd(&bad[0], a);
    if (b) atomicAdd(&bad[1], b);
    if (raw) atomicAdd(&bad[2], raw);
}
__global__ void check_silu(unsigned long long* bad) {
    unsigned long long a = 0, s = 0;
    for (unsigned long long u = blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x; u < (1ull << 32); u += (unsigned long long)gridDim.x * blockDim.x) {
        const float v = __uint_as_float((unsigned)u);
        const float want = v * (1.0f / (1.0f + exp_poly(-v)));
        const float got = silu_poly(v);
        const bool same = __float_as_uint(got) == __float_as_uint(want) || (got != got && want != want);
        a += !same;
        const float ws = 1.0f / (1.0f + exp_poly(-v)), gs = sigmoid_poly(v);
        s += !(__float_as_uint(gs) == __float_as_uint(ws) || (gs != gs && ws != ws));
    }
    if (a) atomicAdd(&bad[3], a);
    if (s) atomicAdd(&bad[4], s);
}
int main() {
    unsigned long long* bad;
    hipMalloc(&bad, 5 * sizeof(*bad));
    hipMemset(bad, 0, 5 * sizeof(*bad));
    hipLaunchKernelGGL(check_recip, dim3(4096), dim3(256), 0, 0, 0x3f800000u, 0x7e800000u, bad);
    hipLaunchKernelGGL(check_silu, dim3(8192), dim3(256), 0, 0, bad);
    unsigned long long h[5];
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("{\"d_range\": \"[1, 2^126], %llu values\", \"mismatches_one_newton_step\": %llu, \"mismatches_two_newton_steps\": %llu, \"mismatches_v_rcp_f32_alone\": %llu, "
           "\"silu_all_2^32_inputs_mismatches\": %llu, \"sigmoid_all_2^32_inputs_mismatches\": %llu}\n",
           (unsigned long long)(0x7e800000u - 0x3f800000u) + 1, h[0], h[1], h[2], h[3], h[4]);
    return (h[3] || h[4]) ? 1 : 0;
}
