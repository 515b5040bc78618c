// oracle/fast.cpp -- AVX2 inner loops of the CPU restatement (TEST INFRASTRUCTURE, see oracle.h).
//
// Two routines whose RESULTS are identical to the plain loops in quant.cpp / gemm.cpp but which run the way lele's x86
// path runs, so that the oracle is usable as the timed CPU baseline of bench.py and full-size layers finish
// in seconds inside the tests:
//
//   orc_fast_int_gemm : sum_k (a - zp_a)(b - zp_b) -- exact integer arithmetic, any evaluation order gives the same bits.
//        Scheme of /root/reference/src/kernels/avx/quantization.rs:1203-1600 (gemm_row_avx2) and 1603- (two-row form):
//        B transposed once to [N][Kpad] with XOR 0x80 (u8 -> i8 = b - 128), A rows as u8; 32 bytes of K per step are
//        widened u8->i16 / i8->i16 and multiplied-accumulated with vpmaddwd into i32 lanes, 8 columns x 2 rows per
//        register block; the zero points are applied through the row sums / column sums:
//            sum (a-za)(b-zb) = sum a*(b-128) + (128 - zb) * rowsum_a - za * (colsum_b - K*zb)
//   orc_fast_sgemm_kord : C[i][j] = fma-chain over k in increasing order (the `acc32` order of gemm.cpp, which is also
//        the order the device's f32 MFMA uses).  lele calls the third-party crate faer here (not in the tree); this is the
//        oracle's own AVX2-FMA kernel and is stated as such wherever it is timed.  4 rows x 16 columns per register block,
//        every element still one k-ordered FMA chain => bit-identical to the scalar loop.
#include <immintrin.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "oracle.h"

namespace {

inline int32_t hsum_epi32(__m256i v) {
    __m128i s = _mm_add_epi32(_mm256_castsi256_si128(v), _mm256_extracti128_si256(v, 1));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0x4e));
    s = _mm_add_epi32(s, _mm_shuffle_epi32(s, 0xb1));
    return _mm_cvtsi128_si32(s);
}

}  // namespace

// a: [m][k] values 0..255 as int32; b: [k][n] values 0..255 as int32 (the layouts quant.cpp already holds).
// acc_out[i*n + j] = sum_k (a[i][k] - zp_a) * (b[k][j] - zp_b) as int32 (wrapping, like the reference's i32 lanes).
extern "C" void orc_fast_int_gemm(const int32_t* a, const int32_t* b, int64_t m, int64_t k, int64_t n, int32_t zp_a,
                                  int32_t zp_b, int32_t* acc_out) {
    const int64_t kp = (k + 31) & ~int64_t(31);
    std::vector<uint8_t> au((size_t)m * kp, 0);
    std::vector<int8_t> bt((size_t)n * kp, 0);  // padding: a = 0 there, so the product is 0 whatever b holds
    std::vector<int32_t> rowsum(m, 0), colsum(n, 0);
    for (int64_t i = 0; i < m; ++i) {
        int32_t s = 0;
        for (int64_t kk = 0; kk < k; ++kk) {
            au[i * kp + kk] = (uint8_t)a[i * k + kk];
            s += a[i * k + kk];
        }
        rowsum[i] = s;
    }
    for (int64_t kk = 0; kk < k; ++kk)
        for (int64_t j = 0; j < n; ++j) {
            const int32_t v = b[kk * n + j];
            bt[j * kp + kk] = (int8_t)(uint8_t)(v ^ 0x80);  // == v - 128
            colsum[j] += v;
        }
    const int32_t c128 = 128 - zp_b;
    auto finish = [&](int64_t i, int64_t j, int32_t dot) {
        // wrapping i32 algebra, as the reference's lanes
        const uint32_t t = (uint32_t)dot + (uint32_t)c128 * (uint32_t)rowsum[i] -
                           (uint32_t)zp_a * ((uint32_t)colsum[j] - (uint32_t)((int32_t)k * zp_b));
        acc_out[i * n + j] = (int32_t)t;
    };
    int64_t i = 0;
    for (; i + 2 <= m; i += 2) {
        const uint8_t *a0 = &au[i * kp], *a1 = a0 + kp;
        int64_t j = 0;
        for (; j + 4 <= n; j += 4) {
            __m256i c00 = _mm256_setzero_si256(), c01 = c00, c02 = c00, c03 = c00, c10 = c00, c11 = c00, c12 = c00, c13 = c00;
            const int8_t *b0 = &bt[j * kp], *b1 = b0 + kp, *b2 = b1 + kp, *b3 = b2 + kp;
            for (int64_t kk = 0; kk < kp; kk += 32) {
                const __m256i fa0 = _mm256_loadu_si256((const __m256i*)(a0 + kk)), fa1 = _mm256_loadu_si256((const __m256i*)(a1 + kk));
                const __m256i a0l = _mm256_cvtepu8_epi16(_mm256_castsi256_si128(fa0)), a0h = _mm256_cvtepu8_epi16(_mm256_extracti128_si256(fa0, 1));
                const __m256i a1l = _mm256_cvtepu8_epi16(_mm256_castsi256_si128(fa1)), a1h = _mm256_cvtepu8_epi16(_mm256_extracti128_si256(fa1, 1));
#define ORC_COL(bp, r0, r1)                                                                                      \
    {                                                                                                            \
        const __m256i fb = _mm256_loadu_si256((const __m256i*)((bp) + kk));                                      \
        const __m256i bl = _mm256_cvtepi8_epi16(_mm256_castsi256_si128(fb)), bh = _mm256_cvtepi8_epi16(_mm256_extracti128_si256(fb, 1)); \
        r0 = _mm256_add_epi32(r0, _mm256_add_epi32(_mm256_madd_epi16(a0l, bl), _mm256_madd_epi16(a0h, bh)));    \
        r1 = _mm256_add_epi32(r1, _mm256_add_epi32(_mm256_madd_epi16(a1l, bl), _mm256_madd_epi16(a1h, bh)));    \
    }
                ORC_COL(b0, c00, c10)
                ORC_COL(b1, c01, c11)
                ORC_COL(b2, c02, c12)
                ORC_COL(b3, c03, c13)
            }
            finish(i, j, hsum_epi32(c00));
            finish(i, j + 1, hsum_epi32(c01));
            finish(i, j + 2, hsum_epi32(c02));
            finish(i, j + 3, hsum_epi32(c03));
            finish(i + 1, j, hsum_epi32(c10));
            finish(i + 1, j + 1, hsum_epi32(c11));
            finish(i + 1, j + 2, hsum_epi32(c12));
            finish(i + 1, j + 3, hsum_epi32(c13));
        }
        for (; j < n; ++j) {
            __m256i c0 = _mm256_setzero_si256(), c1 = c0;
            const int8_t* b0 = &bt[j * kp];
            for (int64_t kk = 0; kk < kp; kk += 32) {
                const __m256i fa0 = _mm256_loadu_si256((const __m256i*)(a0 + kk)), fa1 = _mm256_loadu_si256((const __m256i*)(a1 + kk));
                const __m256i a0l = _mm256_cvtepu8_epi16(_mm256_castsi256_si128(fa0)), a0h = _mm256_cvtepu8_epi16(_mm256_extracti128_si256(fa0, 1));
                const __m256i a1l = _mm256_cvtepu8_epi16(_mm256_castsi256_si128(fa1)), a1h = _mm256_cvtepu8_epi16(_mm256_extracti128_si256(fa1, 1));
                ORC_COL(b0, c0, c1)
            }
            finish(i, j, hsum_epi32(c0));
            finish(i + 1, j, hsum_epi32(c1));
        }
    }
#undef ORC_COL
    for (; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) {
            __m256i c0 = _mm256_setzero_si256();
            const uint8_t* a0 = &au[i * kp];
            const int8_t* b0 = &bt[j * kp];
            for (int64_t kk = 0; kk < kp; kk += 32) {
                const __m256i fa0 = _mm256_loadu_si256((const __m256i*)(a0 + kk));
                const __m256i fb = _mm256_loadu_si256((const __m256i*)(b0 + kk));
                c0 = _mm256_add_epi32(c0, _mm256_add_epi32(
                    _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_castsi256_si128(fa0)), _mm256_cvtepi8_epi16(_mm256_castsi256_si128(fb))),
                    _mm256_madd_epi16(_mm256_cvtepu8_epi16(_mm256_extracti128_si256(fa0, 1)), _mm256_cvtepi8_epi16(_mm256_extracti128_si256(fb, 1)))));
            }
            finish(i, j, hsum_epi32(c0));
        }
}

// C[i][j] = fmaf chain over k = 0..K-1 starting from 0 (A [m][k] row stride lda, B [k][n] row stride ldb, both unit
// column stride).  Identical bits to the scalar `acc32` loop of gemm.cpp::dot.
extern "C" void orc_fast_sgemm_kord(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int64_t k,
                                    int64_t n, float* c, int64_t ldc) {
    int64_t j = 0;
    for (; j + 16 <= n; j += 16) {
        int64_t i = 0;
        for (; i + 4 <= m; i += 4) {
            __m256 c00 = _mm256_setzero_ps(), c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00;
            const float *a0 = a + i * lda, *a1 = a0 + lda, *a2 = a1 + lda, *a3 = a2 + lda;
            for (int64_t kk = 0; kk < k; ++kk) {
                const __m256 b0 = _mm256_loadu_ps(b + kk * ldb + j), b1 = _mm256_loadu_ps(b + kk * ldb + j + 8);
                __m256 av = _mm256_broadcast_ss(a0 + kk);
                c00 = _mm256_fmadd_ps(av, b0, c00);
                c01 = _mm256_fmadd_ps(av, b1, c01);
                av = _mm256_broadcast_ss(a1 + kk);
                c10 = _mm256_fmadd_ps(av, b0, c10);
                c11 = _mm256_fmadd_ps(av, b1, c11);
                av = _mm256_broadcast_ss(a2 + kk);
                c20 = _mm256_fmadd_ps(av, b0, c20);
                c21 = _mm256_fmadd_ps(av, b1, c21);
                av = _mm256_broadcast_ss(a3 + kk);
                c30 = _mm256_fmadd_ps(av, b0, c30);
                c31 = _mm256_fmadd_ps(av, b1, c31);
            }
            _mm256_storeu_ps(c + i * ldc + j, c00);
            _mm256_storeu_ps(c + i * ldc + j + 8, c01);
            _mm256_storeu_ps(c + (i + 1) * ldc + j, c10);
            _mm256_storeu_ps(c + (i + 1) * ldc + j + 8, c11);
            _mm256_storeu_ps(c + (i + 2) * ldc + j, c20);
            _mm256_storeu_ps(c + (i + 2) * ldc + j + 8, c21);
            _mm256_storeu_ps(c + (i + 3) * ldc + j, c30);
            _mm256_storeu_ps(c + (i + 3) * ldc + j + 8, c31);
        }
        for (; i < m; ++i) {
            __m256 c0 = _mm256_setzero_ps(), c1 = c0;
            for (int64_t kk = 0; kk < k; ++kk) {
                const __m256 av = _mm256_broadcast_ss(a + i * lda + kk);
                c0 = _mm256_fmadd_ps(av, _mm256_loadu_ps(b + kk * ldb + j), c0);
                c1 = _mm256_fmadd_ps(av, _mm256_loadu_ps(b + kk * ldb + j + 8), c1);
            }
            _mm256_storeu_ps(c + i * ldc + j, c0);
            _mm256_storeu_ps(c + i * ldc + j + 8, c1);
        }
    }
    for (; j < n; ++j)
        for (int64_t i = 0; i < m; ++i) {
            float acc = 0.0f;
            for (int64_t kk = 0; kk < k; ++kk) acc = __builtin_fmaf(a[i * lda + kk], b[kk * ldb + j], acc);
            c[i * ldc + j] = acc;
        }
}
