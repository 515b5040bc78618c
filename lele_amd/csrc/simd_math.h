// simd_math.h -- device restatement of lele's AVX2 polynomial math (/root/reference/src/kernels/avx/math.rs:11-145).
// Every "8-wide body" element of the x86 kernels goes through these; tails use libm (see eltwise.hip).
// Each operation is the scalar image of one AVX2 instruction, so results are bit-identical to the oracle's intrinsics
// (the library is built with -ffp-contract=off: only the explicit fmaf_ calls fuse).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace lele {

__device__ __forceinline__ float fmaf_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// avx2_exp_ps, avx/math.rs:11-63
__device__ __forceinline__ float exp_poly(float x) {
    x = fmaxf(x, -87.33654f);
    x = fminf(x, 88.72284f);
    const float fx = rintf(x * 1.44269504088896341f);  // _mm256_round_ps(nearest-even)
    x = fmaf_(-fx, 0.693359375f, x);                    // _mm256_fnmadd_ps(fx, ln2_hi, x)
    x = fmaf_(-fx, -2.12194440e-4f, x);
    float y = fmaf_(0.000198712018891638893f, x, 0.00139712726883569741f);
    y = fmaf_(y, x, 0.00833345670066840443f);
    y = fmaf_(y, x, 0.0416657844442129135f);
    y = fmaf_(y, x, 0.166666671633720398f);
    y = fmaf_(y, x, 0.5f);
    y = fmaf_(y, x, 1.0f);
    y = fmaf_(y, x, 1.0f);
    const int e = ((int)fx + 127) << 23;  // cvtps_epi32(fx) is exact: fx is integral
    return y * __int_as_float(e);
}
// 1.0f / d, correctly rounded like the _mm256_div_ps it stands for, for the d = 1 + exp(..) of a sigmoid: v_rcp_f32 (1 ulp) and
// LELE_RECIP_STEPS Newton step (two FMAs) instead of the eleven instructions of the general division (scaling, fix-up), for 1 <= d <= 2^126
// (quotient and residuals are normal numbers there); anything else (d beyond 2^126: the quotient is subnormal; a NaN) takes the
// general division.  Equal to it for EVERY d in the range and the sigmoid / SiLU built on it for every f32 input: tools/recip_check.hip
// checks all of them on the device (profiles/r04_recip_check.json).
#ifndef LELE_RECIP_STEPS
#define LELE_RECIP_STEPS 1
#endif
__device__ __forceinline__ float recip_ge1(float d) {
    if (!(d <= 8.507059173023462e37f)) return 1.0f / d;  // 2^126
    float r = __builtin_amdgcn_rcpf(d);
#pragma unroll
    for (int i = 0; i < LELE_RECIP_STEPS; ++i) r = fmaf_(fmaf_(-d, r, 1.0f), r, r);
    return r;
}
__device__ __forceinline__ float sigmoid_poly(float x) { return recip_ge1(1.0f + exp_poly(-x)); }  // avx/math.rs:66-76
__device__ __forceinline__ float silu_poly(float x) { return x * recip_ge1(1.0f + exp_poly(-x)); }
__device__ __forceinline__ float tanh_poly(float x) {                                            // avx/math.rs:79-97
    const float e = exp_poly(-x * 2.0f);
    const float r = (1.0f - e) / (1.0f + e);
    return copysignf(fabsf(r), x);
}
__device__ __forceinline__ float erf_poly(float x) {  // avx/math.rs:112-145
    const float ax = fabsf(x);
    const float t = 1.0f / fmaf_(0.3275911f, ax, 1.0f);
    float poly = fmaf_(1.061405429f, t, -1.453152027f);
    poly = fmaf_(poly, t, 1.421413741f);
    poly = fmaf_(poly, t, -0.284496736f);
    poly = fmaf_(poly, t, 0.254829592f);
    const float ev = exp_poly(-(ax * ax));
    const float r = fmaf_(-(poly * t), ev, 1.0f);  // _mm256_fnmadd_ps(poly*t, exp, one)
    return __int_as_float(__float_as_int(r) | (__float_as_int(x) & 0x80000000));
}

}  // namespace lele
