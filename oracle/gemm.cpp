// oracle/gemm.cpp -- CPU restatement of lele's f32 GEMM entry points (TEST INFRASTRUCTURE, see oracle.h).
//
//   matmul            /root/reference/src/kernels/gemm.rs:112-222
//   matmul_fused_add  /root/reference/src/kernels/gemm.rs:223-432
//   gemm              /root/reference/src/kernels/gemm.rs:433-535 (x86 branch 477-534)
//
// On x86 lele delegates the inner product to the third-party crate faer 0.24 (`faer::linalg::matmul::matmul`,
// Accum::Replace / Accum::Add, Par::Seq), which is NOT vendored under /root/reference.  Its blocking and
// summation order are therefore unpinned; what IS pinned (shapes, batching/broadcast rules, bias prefill,
// beta*C broadcast cases, alpha, transposes as strides) is restated here around an inner product that is
// accumulated in float64 and rounded once -- the reference every f32 summation order is within a few ulp*K of.
// `acc32 != 0` selects a plain k-ordered float accumulation (one FMA per term) instead, which is the order the
// device's f32 MFMA uses; tests use it to show the device result is bit-identical to that order.
#include <math.h>
#include <string.h>

#include <vector>

#include "oracle.h"

static inline float dot(const float* a, int64_t sa, const float* b, int64_t sb, int64_t k, float init, float alpha,
                        int acc32) {
    if (acc32) {
        float acc = 0.0f;
        for (int64_t i = 0; i < k; ++i) acc = __builtin_fmaf(a[i * sa], b[i * sb], acc);
        return init + alpha * acc;
    }
    double acc = 0.0;
    for (int64_t i = 0; i < k; ++i) acc += (double)a[i * sa] * (double)b[i * sb];
    return (float)((double)init + (double)alpha * acc);
}

extern "C" void orc_fast_sgemm_kord(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int64_t k,
                                    int64_t n, float* c, int64_t ldc);  // fast.cpp: the same k-ordered FMA chains, vectorised
extern "C" int orc_plain_loops;

extern "C" void orc_matmul(const float* a, const float* b, int64_t batch_a, int64_t batch_b, int64_t m, int64_t k,
                           int64_t n, float* out, int acc32) {
    int64_t fb = batch_a > batch_b ? batch_a : batch_b;  // gemm.rs:131
    for (int64_t bi = 0; bi < fb; ++bi) {
        const float* A = a + (batch_a == 1 ? 0 : bi * m * k);  // gemm.rs:156-157
        const float* B = b + (batch_b == 1 ? 0 : bi * k * n);
        float* O = out + bi * m * n;
        if (acc32 && !orc_plain_loops) {  // init 0 + 1.0f * chain == chain exactly (chain + 0.0f keeps every value but -0 -> +0)
            orc_fast_sgemm_kord(A, k, B, n, m, k, n, O, n);
            for (int64_t i = 0; i < m * n; ++i) O[i] = 0.0f + 1.0f * O[i];
            continue;
        }
        for (int64_t i = 0; i < m; ++i)
            for (int64_t j = 0; j < n; ++j) O[i * n + j] = dot(A + i * k, 1, B + j, n, k, 0.0f, 1.0f, acc32);
    }
}

extern "C" void orc_matmul_fused_add(const float* a, const float* b, const float* bias, int64_t bias_len,
                                     int64_t batch_a, int64_t batch_b, int64_t m, int64_t k, int64_t n, float* out,
                                     int acc32) {
    int64_t fb = batch_a > batch_b ? batch_a : batch_b;
    if (bias_len == n) {  // gemm.rs:251-316: rows prefilled with bias, then Accum::Add
        for (int64_t bi = 0; bi < fb; ++bi) {
            const float* A = a + (batch_a == 1 ? 0 : bi * m * k);
            const float* B = b + (batch_b == 1 ? 0 : bi * k * n);
            float* O = out + bi * m * n;
            for (int64_t i = 0; i < m; ++i)
                for (int64_t j = 0; j < n; ++j) O[i * n + j] = dot(A + i * k, 1, B + j, n, k, bias[j], 1.0f, acc32);
        }
        return;
    }
    // gemm.rs:354-415: matmul, then out[i] += bias[i % len] (scalar / full / modulo are all this formula)
    orc_matmul(a, b, batch_a, batch_b, m, k, n, out, acc32);
    int64_t len = fb * m * n;
    for (int64_t i = 0; i < len; ++i) out[i] += bias[i % bias_len];
}

extern "C" void orc_gemm(const float* a, const float* b, const float* c, int64_t c_len, float alpha, float beta,
                         int trans_a, int trans_b, int64_t m, int64_t k, int64_t n, float* out, int acc32) {
    // C prefill, gemm.rs:484-515
    std::vector<float> pre(m * n, 0.0f);
    if (c && beta != 0.0f) {
        for (int64_t i = 0; i < m; ++i)
            for (int64_t j = 0; j < n; ++j) {
                float v;
                if (c_len == m * n)
                    v = c[i * n + j] * beta;
                else if (c_len == n)
                    v = c[j] * beta;
                else if (c_len == m)
                    v = c[i] * beta;
                else if (c_len == 1)
                    v = c[0] * beta;
                else
                    v = c[(i * n + j) % c_len] * beta;
                pre[i * n + j] = v;
            }
    }
    // strides, gemm.rs:517-520
    int64_t rsa = trans_a ? 1 : k, csa = trans_a ? m : 1;
    int64_t rsb = trans_b ? 1 : n, csb = trans_b ? k : 1;
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j)
            out[i * n + j] = dot(a + i * rsa, csa, b + j * csb, rsb, k, pre[i * n + j], alpha, acc32);
}
