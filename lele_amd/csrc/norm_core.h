// norm_core.h -- row reductions shared by the normalisation kernels (eltwise.hip) and the fused attention kernel
// (attention.hip): the 32 lanes of a row play the four 8-wide AVX accumulators of lele's x86 kernels
// (/root/reference/src/kernels/avx/norm.rs:10-229), so sums are formed in the reference's order and come out bit-identical.
#pragma once
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
    float s01 = s + __shfl_down(s, 8, 32), q01 = q + __shfl_down(q, 8, 32);
    float sv = s01 + __shfl_down(s01, 16, 32), qv = q01 + __shfl_down(q01, 16, 32);
    float last = 0.0f;  // the partially filled register row v[nfull]
#pragma unroll
    for (int c = 0; c < NT; ++c)
        if (c == nfull) last = v[c];
    const int rem = n - 32 * nfull, nch = rem >> 3;
#pragma unroll
    for (int r = 0; r < 3; ++r)
        if (r < nch) {
            const float got = __shfl(last, 8 * r + (l & 7), 32);
            if (l < 8) {
                if (PLAIN) sv = sv + got;
                if (SQUARE) qv = fmaf_(got, got, qv);
            }
        }
    sv = sv + __shfl_down(sv, 4, 32);
    qv = qv + __shfl_down(qv, 4, 32);
    sv = sv + __shfl_down(sv, 2, 32);
    qv = qv + __shfl_down(qv, 2, 32);
    sv = sv + __shfl_down(sv, 1, 32);
    qv = qv + __shfl_down(qv, 1, 32);
#pragma unroll
    for (int t = 0; t < 7; ++t)
        if (t < (rem & 7)) {
            const float got = __shfl(last, 8 * nch + t, 32);
            if (PLAIN) sv = sv + got;
            if (SQUARE) qv = qv + got * got;
        }
    *out_sum = __shfl(sv, 0, 32);
    *out_sq = __shfl(qv, 0, 32);
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
    for (int off = 16; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 32));
    const int body = len & ~7;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = 32 * c + l;
        v[c] = j < body ? exp_poly(v[c] - m) : expf(v[c] - m);
    }
    float sum, dummy;
    row_sums_reg<NT, false, true>(v, len, l, &sum, &dummy);
    const float inv_sum = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < NT; ++c) v[c] = v[c] * inv_sum;
}

}  // namespace lele
