// oracle/quant.cpp -- CPU restatement of lele's dynamic-quantised u8 path (TEST INFRASTRUCTURE, see oracle.h).
//
//   fused_quantized_linear            /root/reference/src/kernels/quantization.rs:77-169 (x86 branch 86-128)
//     -> fused_dq_gemm_avx2           /root/reference/src/kernels/avx/quantization.rs:225-417
//     -> dq_to_u8_rowsums_avx2        /root/reference/src/kernels/avx/quantization.rs:102-219
//     -> gemm_row_avx2 / gemm_2rows   /root/reference/src/kernels/avx/quantization.rs:1203-1600, 1603-
//   dynamic_quantize_linear           /root/reference/src/kernels/quantization.rs:1628-1657
//     -> dynamic_quantize_linear_avx2 /root/reference/src/kernels/avx/quantization.rs:832-927
//   mat_mul_integer{,_with_bias,_with_scale_bias,_relu}
//                                     /root/reference/src/kernels/quantization.rs:8-72, 927-992
//     -> mat_mul_integer_fused_f32_avx2 /root/reference/src/kernels/avx/quantization.rs:642-830
//
// The AVX2 code computes  sum_k a*(b^0x80) + rowsum*(128-zp_b) - zp_a*(colsum - K*zp_b)  in wrapping i32, which is
// exactly  sum_k (a - zp_a)*(b - zp_b); that integer is restated directly.  Everything that rounds is kept
// operation-for-operation: the quantiser is round-half-even(fma(x, 1/scale, zp)) inside the 8-wide SIMD body and
// f32::round(x*inv + zp) (half away from zero, two roundings) in the scalar tail; the epilogue is
// (float)acc * combined_scale, then + bias (separate mul and add), then max(.,0).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "oracle.h"

namespace {

struct DynQ {
    float scale, zp, inv_scale;
};

// min/max + scale/zero-point, avx/quantization.rs:110-140 (== 855-884)
DynQ dyn_params(const float* x, int64_t len) {
    float min_val = 3.40282347e+38f, max_val = -3.40282347e+38f;  // f32::MAX / f32::MIN seeds
    for (int64_t i = 0; i < len; ++i) {
        if (x[i] < min_val) min_val = x[i];
        if (x[i] > max_val) max_val = x[i];
    }
    float adjusted_max = max_val > 0.0f ? max_val : 0.0f;
    float adjusted_min = min_val < 0.0f ? min_val : 0.0f;
    float range = adjusted_max - adjusted_min;
    if (!(range > 1e-5f)) range = 1e-5f;
    DynQ q;
    q.scale = range / 255.0f;
    float z = roundf(-adjusted_min / q.scale);  // f32::round: half away from zero
    q.zp = z < 0.0f ? 0.0f : (z > 255.0f ? 255.0f : z);
    q.inv_scale = 1.0f / q.scale;
    return q;
}

inline float quant_simd(float v, const DynQ& q) {  // _mm256_fmadd_ps + _mm256_round_ps(nearest-even) + clamp
    float r = nearbyintf(__builtin_fmaf(v, q.inv_scale, q.zp));
    return r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
}
inline float quant_tail(float v, const DynQ& q) {  // (v * inv_scale + zp).round().clamp(0, 255)
    float r = roundf(v * q.inv_scale + q.zp);
    return r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
}
inline int32_t f32_to_u8(float v) {  // cvtps_epi32 (nearest even) + packs/packus saturation
    float r = nearbyintf(v);
    if (r < 0.0f) return 0;
    if (r > 255.0f) return 255;
    return (int32_t)r;
}

extern "C" void orc_fast_int_gemm(const int32_t* a, const int32_t* b, int64_t m, int64_t k, int64_t n, int32_t zp_a,
                                  int32_t zp_b, int32_t* acc_out);  // fast.cpp: the reference's vpmaddwd scheme, same integers
extern "C" {
int orc_plain_loops = 0;
}  // tests set this to 1 to run the plain triple loop below instead (cross-check)

// out[i][j] = epilogue( sum_k (a[i][k]-zp_a)*(b[k][j]-zp_b) )
void int_gemm_epilogue(const int32_t* a, const int32_t* b, int64_t m, int64_t k, int64_t n, int32_t zp_a, int32_t zp_b,
                       const float* scale, int64_t scale_len, const float* bias, int relu, float* out) {
    std::vector<int32_t> accs;
    if (!orc_plain_loops && m * n > 0) {
        accs.resize(m * n);
        orc_fast_int_gemm(a, b, m, k, n, zp_a, zp_b, accs.data());
    }
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) {
            int32_t total;
            if (accs.empty()) {
                int64_t acc = 0;
                for (int64_t kk = 0; kk < k; ++kk) acc += (int64_t)(a[i * k + kk] - zp_a) * (int64_t)(b[kk * n + j] - zp_b);
                total = (int32_t)acc;
            } else {
                total = accs[i * n + j];
            }
            float vf = (float)total;  // _mm256_cvtepi32_ps
            if (scale) vf = vf * (scale_len == 1 ? scale[0] : scale[j]);
            if (bias) vf = vf + bias[j];
            if (relu && !(vf > 0.0f)) vf = 0.0f;  // _mm256_max_ps(vf, 0)
            out[i * n + j] = vf;
        }
}

}  // namespace

extern "C" void orc_dynamic_quantize_linear(const float* x, int64_t len, float* y, float* scale, float* zp) {
    if (orc_scalar_mode() && len > 0) return orc_scalar_dynamic_quantize_linear(x, len, y, scale, zp);
    if (len == 0) {  // avx/quantization.rs:845-851
        *scale = 1.0f;
        *zp = 0.0f;
        return;
    }
    DynQ q = dyn_params(x, len);
    *scale = q.scale;
    *zp = q.zp;
    int64_t simd_end = (len / 8) * 8;
    for (int64_t i = 0; i < simd_end; ++i) y[i] = quant_simd(x[i], q);
    for (int64_t i = simd_end; i < len; ++i) y[i] = quant_tail(x[i], q);
}

extern "C" void orc_fused_quantized_linear(const float* input, int64_t batch, int64_t m, int64_t k, int64_t n,
                                           const float* weight /*[k,n] f32-encoded u8*/, const float* weight_scale,
                                           int64_t weight_scale_len, float weight_zero, const float* bias /*or NULL*/,
                                           int relu, float* out) {
    const int32_t zp_b = (int32_t)weight_zero;  // `v as i32`
    std::vector<int32_t> bq(k * n), aq(m * k);
    for (int64_t i = 0; i < k * n; ++i) bq[i] = f32_to_u8(weight[i]);  // transpose_b_from_f32_avx2
    std::vector<float> comb(weight_scale_len > 1 ? weight_scale_len : 1);
    for (int64_t bi = 0; bi < batch; ++bi) {  // quantization.rs:104-128: one dynamic range PER BATCH SLICE
        const float* x = input + bi * m * k;
        DynQ q = dyn_params(x, m * k);
        for (int64_t r = 0; r < m; ++r) {
            int64_t simd_k = (k / 8) * 8;  // 16-wide then 8-wide SIMD body, scalar remainder per ROW
            for (int64_t kk = 0; kk < simd_k; ++kk) aq[r * k + kk] = (int32_t)quant_simd(x[r * k + kk], q);
            for (int64_t kk = simd_k; kk < k; ++kk) aq[r * k + kk] = (int32_t)quant_tail(x[r * k + kk], q);
        }
        const int32_t zp_a = (int32_t)q.zp;
        if (weight_scale_len <= 1)
            comb[0] = q.scale * weight_scale[0];
        else
            for (int64_t j = 0; j < weight_scale_len; ++j) comb[j] = q.scale * weight_scale[j];
        int_gemm_epilogue(aq.data(), bq.data(), m, k, n, zp_a, zp_b, comb.data(), weight_scale_len <= 1 ? 1 : n, bias,
                          relu, out + bi * m * n);
    }
}

extern "C" void orc_mat_mul_integer(const float* a, const float* b, int64_t batch_a, int64_t batch_b, int64_t m,
                                    int64_t k, int64_t n, float zp_a_f, float zp_b_f, const float* scale,
                                    int64_t scale_len, const float* bias, int relu, float* out) {
    const int32_t zp_a = (int32_t)zp_a_f, zp_b = (int32_t)zp_b_f;
    int64_t fb = batch_a > batch_b ? batch_a : batch_b;
    std::vector<int32_t> aq(m * k), bq(k * n);
    for (int64_t bi = 0; bi < fb; ++bi) {
        const float* A = a + (batch_a == 1 ? 0 : bi * m * k);
        const float* B = b + (batch_b == 1 ? 0 : bi * k * n);
        for (int64_t i = 0; i < m * k; ++i) aq[i] = f32_to_u8(A[i]);
        for (int64_t i = 0; i < k * n; ++i) bq[i] = f32_to_u8(B[i]);
        int_gemm_epilogue(aq.data(), bq.data(), m, k, n, zp_a, zp_b, scale, scale_len, bias, relu, out + bi * m * n);
    }
}
