// norm_core.h -- row reductions shared by the normalisation kernels (eltwise.hip) and the fused attention kernel
// (attention.hip): the 32 lanes of a row play the four 8-wide AVX accumulators of lele's x86 kernels
// (/root/reference/src/kernels/avx/norm.rs:10-229), so sums are formed in the reference's order and come out bit-identical.
#pragma once
#include "lane_ops.h"
#include "simd_math.h"

namespace lele {

// Register-resident form of row_sums for rows of at most 32*NT elements: lane l holds v[c] = row[32c + l].  The same
// additions in the same order as row_sums (so the same bits); elements another lane owns (the 8-wide remainder chunks
// and the scalar tail) arrive by __shfl instead of a second trip to memory.
template <int NT, bool SQUARE, bool PLAIN>
__device__ __forceinline__ void row_sums_reg(const float (&v)[NT], int n, int l, float* out_sum, float* out_sq) {
    float s = 0.0f, q = 0.0f;
    const int nfull = n >> 5;
#pragma unroll
    for (int c = 0; c < NT; ++c)
        if (c < nfull) {
            if (PLAIN) s = s + v[c];
            if (SQUARE) q = fmaf_(v[c], v[c], q);
        }
    // cross-lane moves on the DPP / permlane network (lane_ops.h): the same additions in the same order as the __shfl form, whose
    // ~15 dependent ds_bpermute round trips per row were the cost of this routine.  Only lanes 0..7 of the group carry the sum from
    // here on (then lanes 0..3, 0..1, 0), which is where each move's result is defined.
    float s01 = s + row_down<8>(s), q01 = q + row_down<8>(q);
    float sv = s01 + swap16(s01), qv = q01 + swap16(q01);
    float last = 0.0f;  // the partially filled register row v[nfull]
#pragma unroll
    for (int c = 0; c < NT; ++c)
        if (c == nfull) last = v[c];
    const int rem = n - 32 * nfull, nch = rem >> 3;
    // 8-wide remainder chunk r lives in lanes 8r .. 8r + 7; lane l < 8 adds element l of it
    if (nch > 0 && l < 8) {
        if (PLAIN) sv = sv + last;
        if (SQUARE) qv = fmaf_(last, last, qv);
    }
    if (nch > 1) {
        const float got = row_down<8>(last);
        if (l < 8) {
            if (PLAIN) sv = sv + got;
            if (SQUARE) qv = fmaf_(got, got, qv);
        }
    }
    if (nch > 2) {
        const float got = swap16(last);
        if (l < 8) {
            if (PLAIN) sv = sv + got;
            if (SQUARE) qv = fmaf_(got, got, qv);
        }
    }
    sv = sv + row_down<4>(sv);
    qv = qv + row_down<4>(qv);
    sv = sv + row_down<2>(sv);
    qv = qv + row_down<2>(qv);
    sv = sv + row_down<1>(sv);
    qv = qv + row_down<1>(qv);
#pragma unroll
    for (int t = 0; t < 7; ++t)
        if (t < (rem & 7)) {  // uniform; only lane 0's running sum is read below
            const float got = group_read(last, 8 * nch + t);
            if (PLAIN) sv = sv + got;
            if (SQUARE) qv = qv + got * got;
        }
    *out_sum = group_read(sv, 0);
    *out_sq = group_read(qv, 0);
}

// One row of softmax_reg_kernel (eltwise.hip; avx/norm.rs:139-229) on values already in registers: lane l of a 32-lane group
// holds v[c] = row[32c + l] (already multiplied by the scale when there is one); on return v[c] holds the probabilities.
// The same operations in the same order as the stand-alone kernel, so the same bits.
template <int NT>
__device__ __forceinline__ void softmax_row_reg(float (&v)[NT], int len, int l) {
    float m = -3.40282347e+38f;
#pragma unroll
    for (int c = 0; c < NT; ++c)
        if (32 * c + l < len) m = fmaxf(m, v[c]);
    m = group_max32(m);
    const int body = len & ~7;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = 32 * c + l;
        // register rows entirely inside the 8-wide body (all but one) skip the libm call: a per-lane select would evaluate both
        // functions for every element (the condition is uniform: `body` depends on the row length only)
        if (32 * c + 32 <= body) v[c] = exp_poly(v[c] - m);
        else v[c] = j < body ? exp_poly(v[c] - m) : expf(v[c] - m);
    }
    float sum, dummy;
    row_sums_reg<NT, false, true>(v, len, l, &sum, &dummy);
    const float inv_sum = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < NT; ++c) v[c] = v[c] * inv_sum;
}

// The same row as softmax_row_reg with the chip's own exponential and tree reductions: exp(x) = v_exp_f32(x * log2 e), the sum over
// the 32-lane group on the DPP / permlane network, one reciprocal.  ~60 vector instructions per row of 171 against ~670 for the
// replica of lele's polynomial / libm-tail / 4 x 8-accumulator routine (which cost the one-launch attention kernel 36 % of its
// life).  Not the reference's bits: the probabilities agree with avx/norm.rs:139-229 to ~1e-6 relative (v_exp_f32 is good to
// 1 ulp; the argument's rounding costs |x| * 6e-8, and a term of size |x| carries exp(-|x|) of the row's weight).  Used only
// INSIDE lele_hip_attention_view, whose products already differ from the node sequence by summation order; the stand-alone
// softmax operator stays bit-exact.
template <int NT>
__device__ __forceinline__ void softmax_row_fast(float (&v)[NT], int len, int l) {
    float m = -3.40282347e+38f;
#pragma unroll
    for (int c = 0; c < NT; ++c)
        if (32 * c + l < len) m = fmaxf(m, v[c]);
    m = group_max32(m);
    float sum = 0.0f;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        v[c] = 32 * c + l < len ? __builtin_amdgcn_exp2f((v[c] - m) * 1.44269504088896341f) : 0.0f;
        sum += v[c];
    }
    sum = group_allreduce32(sum, [](float a, float b) { return a + b; });
    const float inv_sum = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < NT; ++c) v[c] = v[c] * inv_sum;
}

}  // namespace lele
