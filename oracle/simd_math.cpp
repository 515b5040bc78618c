// oracle/simd_math.cpp -- CPU restatement of lele's x86 AVX2 activation / normalisation kernels, written with the
// SAME <immintrin.h> intrinsic sequences (TEST INFRASTRUCTURE, see oracle.h).
//
//   avx2_exp_ps / sigmoid / tanh / silu / erf      /root/reference/src/kernels/avx/math.rs:11-145
//   buffer kernels (8-wide body + libm scalar tail) /root/reference/src/kernels/avx/math.rs:232-603
//   layer_norm_x86                                  /root/reference/src/kernels/avx/norm.rs:10-133
//   softmax                                         /root/reference/src/kernels/avx/norm.rs:139-229
//   rms_norm_x86                                    /root/reference/src/kernels/avx/norm.rs:236-307
//   batch_norm (+ batch_norm_spatial_x86)           /root/reference/src/kernels/norm.rs:313-418, avx/norm.rs:313-345
#include <immintrin.h>
#include <math.h>
#include <stdint.h>

#include "oracle.h"

static inline __m256 exp_ps(__m256 x) {  // avx/math.rs:11-63
    const __m256 log2ef = _mm256_set1_ps(1.44269504088896341f), ln2_hi = _mm256_set1_ps(0.693359375f),
                 ln2_lo = _mm256_set1_ps(-2.12194440e-4f), c1 = _mm256_set1_ps(0.5f),
                 c2 = _mm256_set1_ps(0.166666671633720398f), c3 = _mm256_set1_ps(0.0416657844442129135f),
                 c4 = _mm256_set1_ps(0.00833345670066840443f), c5 = _mm256_set1_ps(0.00139712726883569741f),
                 c6 = _mm256_set1_ps(0.000198712018891638893f), one = _mm256_set1_ps(1.0f);
    x = _mm256_max_ps(x, _mm256_set1_ps(-87.33654f));
    x = _mm256_min_ps(x, _mm256_set1_ps(88.72284f));
    __m256 fx = _mm256_round_ps(_mm256_mul_ps(x, log2ef), _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC);
    x = _mm256_fnmadd_ps(fx, ln2_hi, x);
    x = _mm256_fnmadd_ps(fx, ln2_lo, x);
    __m256 y = _mm256_fmadd_ps(c6, x, c5);
    y = _mm256_fmadd_ps(y, x, c4);
    y = _mm256_fmadd_ps(y, x, c3);
    y = _mm256_fmadd_ps(y, x, c2);
    y = _mm256_fmadd_ps(y, x, c1);
    y = _mm256_fmadd_ps(y, x, one);
    y = _mm256_fmadd_ps(y, x, one);
    __m256i e = _mm256_cvtps_epi32(fx);
    e = _mm256_add_epi32(e, _mm256_set1_epi32(127));
    e = _mm256_slli_epi32(e, 23);
    return _mm256_mul_ps(y, _mm256_castsi256_ps(e));
}
static inline __m256 sigmoid_ps(__m256 x) {  // avx/math.rs:66-76
    const __m256 one = _mm256_set1_ps(1.0f);
    __m256 e = exp_ps(_mm256_xor_ps(x, _mm256_set1_ps(-0.0f)));
    return _mm256_div_ps(one, _mm256_add_ps(one, e));
}
static inline __m256 tanh_ps(__m256 x) {  // avx/math.rs:79-97
    const __m256 one = _mm256_set1_ps(1.0f), two = _mm256_set1_ps(2.0f), sm = _mm256_set1_ps(-0.0f);
    __m256 e = exp_ps(_mm256_mul_ps(_mm256_xor_ps(x, sm), two));
    __m256 r = _mm256_div_ps(_mm256_sub_ps(one, e), _mm256_add_ps(one, e));
    return _mm256_or_ps(_mm256_andnot_ps(sm, r), _mm256_and_ps(x, sm));
}
static inline __m256 silu_ps(__m256 x) { return _mm256_mul_ps(x, sigmoid_ps(x)); }  // avx/math.rs:100-106
static inline __m256 erf_ps(__m256 x) {  // avx/math.rs:112-145
    const __m256 sm = _mm256_set1_ps(-0.0f), one = _mm256_set1_ps(1.0f), p = _mm256_set1_ps(0.3275911f),
                 a1 = _mm256_set1_ps(0.254829592f), a2 = _mm256_set1_ps(-0.284496736f),
                 a3 = _mm256_set1_ps(1.421413741f), a4 = _mm256_set1_ps(-1.453152027f),
                 a5 = _mm256_set1_ps(1.061405429f);
    __m256 sign = _mm256_and_ps(x, sm), ax = _mm256_andnot_ps(sm, x);
    __m256 t = _mm256_div_ps(one, _mm256_fmadd_ps(p, ax, one));
    __m256 poly = _mm256_fmadd_ps(a5, t, a4);
    poly = _mm256_fmadd_ps(poly, t, a3);
    poly = _mm256_fmadd_ps(poly, t, a2);
    poly = _mm256_fmadd_ps(poly, t, a1);
    __m256 ev = exp_ps(_mm256_xor_ps(_mm256_mul_ps(ax, ax), sm));
    __m256 r = _mm256_fnmadd_ps(_mm256_mul_ps(poly, t), ev, one);
    return _mm256_or_ps(r, sign);
}
static inline float hsum(__m256 v) {  // avx/math.rs:151-160
    __m128 s = _mm_add_ps(_mm256_castps256_ps128(v), _mm256_extractf128_ps(v, 1));
    s = _mm_add_ps(s, _mm_movehl_ps(s, s));
    s = _mm_add_ss(s, _mm_shuffle_ps(s, s, 1));
    return _mm_cvtss_f32(s);
}

// op ids shared with tests: 0 exp 1 sigmoid 2 tanh 3 silu 4 erf 5 gelu 6 fast_gelu 7 relu 8 sqrt
extern "C" void orc_unary_simd(int op, const float* in, float* out, int64_t len) {
    int64_t i = 0;
    const __m256 half = _mm256_set1_ps(0.5f), one = _mm256_set1_ps(1.0f);
    // scalar mode (math.rs:927-945, 975-, 1028-, ...: the cfg(not) bodies call libm per element): the loop below IS the tail loop
    for (; !orc_scalar_mode() && i + 8 <= len; i += 8) {
        __m256 x = _mm256_loadu_ps(in + i), r;
        switch (op) {
            case 0: r = exp_ps(x); break;
            case 1: r = sigmoid_ps(x); break;
            case 2: r = tanh_ps(x); break;
            case 3: r = silu_ps(x); break;
            case 4: r = erf_ps(x); break;
            case 5:  // gelu_kernel, avx/math.rs:530-560
                r = _mm256_mul_ps(_mm256_mul_ps(x, half),
                                  _mm256_add_ps(one, erf_ps(_mm256_mul_ps(x, _mm256_set1_ps(0.7071067811865475f)))));
                break;
            case 6: {  // fast_gelu_kernel, avx/math.rs:567-602
                __m256 x3 = _mm256_mul_ps(_mm256_mul_ps(x, x), x);
                __m256 inner = _mm256_mul_ps(_mm256_set1_ps(0.7978845608028654f),
                                             _mm256_fmadd_ps(_mm256_set1_ps(0.044715f), x3, x));
                r = _mm256_mul_ps(_mm256_mul_ps(x, half), _mm256_add_ps(one, tanh_ps(inner)));
                break;
            }
            case 7: r = _mm256_max_ps(x, _mm256_setzero_ps()); break;
            default: r = _mm256_sqrt_ps(x); break;
        }
        _mm256_storeu_ps(out + i, r);
    }
    for (; i < len; ++i) {  // scalar tails call libm (avx/math.rs:300-304, 320-323, 340-344, 360-364, ...)
        float x = in[i];
        switch (op) {
            case 0: out[i] = expf(x); break;
            case 1: out[i] = 1.0f / (1.0f + expf(-x)); break;
            case 2: out[i] = tanhf(x); break;
            case 3: out[i] = x / (1.0f + expf(-x)); break;
            case 4: out[i] = erff(x); break;
            case 5: out[i] = x * 0.5f * (1.0f + erff(x * 0.7071067811865475f)); break;
            case 6: {
                float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
                out[i] = 0.5f * x * (1.0f + tanhf(inner));
                break;
            }
            case 7: out[i] = x > 0.0f ? x : 0.0f; break;
            default: out[i] = sqrtf(x); break;
        }
    }
}

extern "C" void orc_layer_norm(const float* input, const float* scale, const float* bias, float* output,
                               int64_t norm_size, int64_t outer_size, float epsilon) {  // avx/norm.rs:10-133
    if (orc_scalar_mode()) return orc_scalar_layer_norm(input, scale, bias, output, norm_size, outer_size, epsilon);
    float inv_n = 1.0f / (float)norm_size;
    for (int64_t i = 0; i < outer_size; ++i) {
        const float* in = input + i * norm_size;
        float* out = output + i * norm_size;
        __m256 s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0, q0 = s0, q1 = s0, q2 = s0, q3 = s0;
        int64_t j = 0;
        for (; j + 32 <= norm_size; j += 32) {
            __m256 v0 = _mm256_loadu_ps(in + j), v1 = _mm256_loadu_ps(in + j + 8), v2 = _mm256_loadu_ps(in + j + 16),
                   v3 = _mm256_loadu_ps(in + j + 24);
            s0 = _mm256_add_ps(s0, v0);
            s1 = _mm256_add_ps(s1, v1);
            s2 = _mm256_add_ps(s2, v2);
            s3 = _mm256_add_ps(s3, v3);
            q0 = _mm256_fmadd_ps(v0, v0, q0);
            q1 = _mm256_fmadd_ps(v1, v1, q1);
            q2 = _mm256_fmadd_ps(v2, v2, q2);
            q3 = _mm256_fmadd_ps(v3, v3, q3);
        }
        __m256 sv = _mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3));
        __m256 qv = _mm256_add_ps(_mm256_add_ps(q0, q1), _mm256_add_ps(q2, q3));
        for (; j + 8 <= norm_size; j += 8) {
            __m256 v = _mm256_loadu_ps(in + j);
            sv = _mm256_add_ps(sv, v);
            qv = _mm256_fmadd_ps(v, v, qv);
        }
        float sum = hsum(sv), sumsq = hsum(qv);
        for (; j < norm_size; ++j) {
            float v = in[j];
            sum += v;
            sumsq += v * v;
        }
        float mean = sum * inv_n;
        float var = sumsq * inv_n - mean * mean;
        float inv_std = 1.0f / sqrtf(var + epsilon);
        __m256 mv = _mm256_set1_ps(mean), iv = _mm256_set1_ps(inv_std);
        for (j = 0; j + 8 <= norm_size; j += 8) {
            __m256 v = _mm256_loadu_ps(in + j);
            __m256 r = _mm256_fmadd_ps(_mm256_mul_ps(_mm256_sub_ps(v, mv), iv), _mm256_loadu_ps(scale + j),
                                       _mm256_loadu_ps(bias + j));
            _mm256_storeu_ps(out + j, r);
        }
        for (; j < norm_size; ++j) out[j] = (in[j] - mean) * inv_std * scale[j] + bias[j];
    }
}

extern "C" void orc_softmax_lastdim(const float* input, float* output, int64_t outer, int64_t len) {  // avx/norm.rs:139-229
    if (orc_scalar_mode()) return orc_scalar_softmax_lastdim(input, output, outer, len);
    for (int64_t r = 0; r < outer; ++r) {
        const float* src = input + r * len;
        float* dst = output + r * len;
        __m256 m0 = _mm256_set1_ps(-3.40282347e+38f), m1 = m0, m2 = m0, m3 = m0;
        int64_t j = 0;
        for (; j + 32 <= len; j += 32) {
            m0 = _mm256_max_ps(m0, _mm256_loadu_ps(src + j));
            m1 = _mm256_max_ps(m1, _mm256_loadu_ps(src + j + 8));
            m2 = _mm256_max_ps(m2, _mm256_loadu_ps(src + j + 16));
            m3 = _mm256_max_ps(m3, _mm256_loadu_ps(src + j + 24));
        }
        __m256 mv = _mm256_max_ps(_mm256_max_ps(m0, m1), _mm256_max_ps(m2, m3));
        for (; j + 8 <= len; j += 8) mv = _mm256_max_ps(mv, _mm256_loadu_ps(src + j));
        __m128 m128 = _mm_max_ps(_mm256_castps256_ps128(mv), _mm256_extractf128_ps(mv, 1));
        __m128 m64 = _mm_max_ps(m128, _mm_movehl_ps(m128, m128));
        float max_val = _mm_cvtss_f32(_mm_max_ss(m64, _mm_shuffle_ps(m64, m64, 1)));
        for (int64_t k = j; k < len; ++k) max_val = fmaxf(max_val, src[k]);
        __m256 mb = _mm256_set1_ps(max_val), s0 = _mm256_setzero_ps(), s1 = s0, s2 = s0, s3 = s0;
        for (j = 0; j + 32 <= len; j += 32) {
            __m256 e0 = exp_ps(_mm256_sub_ps(_mm256_loadu_ps(src + j), mb));
            __m256 e1 = exp_ps(_mm256_sub_ps(_mm256_loadu_ps(src + j + 8), mb));
            __m256 e2 = exp_ps(_mm256_sub_ps(_mm256_loadu_ps(src + j + 16), mb));
            __m256 e3 = exp_ps(_mm256_sub_ps(_mm256_loadu_ps(src + j + 24), mb));
            _mm256_storeu_ps(dst + j, e0);
            _mm256_storeu_ps(dst + j + 8, e1);
            _mm256_storeu_ps(dst + j + 16, e2);
            _mm256_storeu_ps(dst + j + 24, e3);
            s0 = _mm256_add_ps(s0, e0);
            s1 = _mm256_add_ps(s1, e1);
            s2 = _mm256_add_ps(s2, e2);
            s3 = _mm256_add_ps(s3, e3);
        }
        __m256 sv = _mm256_add_ps(_mm256_add_ps(s0, s1), _mm256_add_ps(s2, s3));
        for (; j + 8 <= len; j += 8) {
            __m256 e = exp_ps(_mm256_sub_ps(_mm256_loadu_ps(src + j), mb));
            _mm256_storeu_ps(dst + j, e);
            sv = _mm256_add_ps(sv, e);
        }
        float sum = hsum(sv);
        for (int64_t k = j; k < len; ++k) {
            float e = expf(src[k] - max_val);
            dst[k] = e;
            sum += e;
        }
        float inv_sum = 1.0f / sum;
        for (int64_t k = 0; k < len; ++k) dst[k] = dst[k] * inv_sum;
    }
}

extern "C" void orc_rms_norm(const float* input, const float* weight, float* output, int64_t norm_size,
                             int64_t outer_size, float epsilon) {  // avx/norm.rs:236-307
    float inv_n = 1.0f / (float)norm_size;
    for (int64_t i = 0; i < outer_size; ++i) {
        const float* in = input + i * norm_size;
        float* out = output + i * norm_size;
        __m256 q0 = _mm256_setzero_ps(), q1 = q0, q2 = q0, q3 = q0;
        int64_t j = 0;
        for (; j + 32 <= norm_size; j += 32) {
            __m256 v0 = _mm256_loadu_ps(in + j), v1 = _mm256_loadu_ps(in + j + 8), v2 = _mm256_loadu_ps(in + j + 16),
                   v3 = _mm256_loadu_ps(in + j + 24);
            q0 = _mm256_fmadd_ps(v0, v0, q0);
            q1 = _mm256_fmadd_ps(v1, v1, q1);
            q2 = _mm256_fmadd_ps(v2, v2, q2);
            q3 = _mm256_fmadd_ps(v3, v3, q3);
        }
        q0 = _mm256_add_ps(_mm256_add_ps(q0, q1), _mm256_add_ps(q2, q3));
        for (; j + 8 <= norm_size; j += 8) {
            __m256 v = _mm256_loadu_ps(in + j);
            q0 = _mm256_fmadd_ps(v, v, q0);
        }
        float sumsq = hsum(q0);
        for (; j < norm_size; ++j) sumsq += in[j] * in[j];
        float rms_inv = 1.0f / sqrtf(sumsq * inv_n + epsilon);
        __m256 rv = _mm256_set1_ps(rms_inv);
        for (j = 0; j + 8 <= norm_size; j += 8) {
            __m256 w = _mm256_mul_ps(_mm256_loadu_ps(weight + j), rv);
            _mm256_storeu_ps(out + j, _mm256_mul_ps(_mm256_loadu_ps(in + j), w));
        }
        for (; j < norm_size; ++j) out[j] = in[j] * rms_inv * weight[j];
    }
}

extern "C" void orc_batch_norm(const float* src, const float* s, const float* b, const float* m, const float* v,
                               float epsilon, int64_t outer, int64_t c, int64_t inner, float* out) {
    for (int64_t i = 0; i < outer; ++i)
        for (int64_t j = 0; j < c; ++j) {
            float scale_val = s[j] / sqrtf(v[j] + epsilon);  // norm.rs:339-340
            float bias_val = b[j] - m[j] * scale_val;
            const float* p = src + (i * c + j) * inner;
            float* o = out + (i * c + j) * inner;
            __m256 sv = _mm256_set1_ps(scale_val), bv = _mm256_set1_ps(bias_val);
            int64_t k = 0;
            for (; k + 8 <= inner; k += 8) _mm256_storeu_ps(o + k, _mm256_fmadd_ps(_mm256_loadu_ps(p + k), sv, bv));
            for (; k < inner; ++k) o[k] = p[k] * scale_val + bias_val;
        }
}

// ---------------------------------------------------------------------------------------------------------------
// LSTM / GRU (batch 1, one direction), /root/reference/src/kernels/rnn.rs:15-432.  The two GEMVs per step go through
// faer in the reference (summation order unpinned): here they are float64-accumulated dot products rounded once.
// The gate arithmetic is the reference's AVX2 code (poly sigmoid / tanh in the 8-wide body, libm in the tail).
static inline float dot64(const float* a, const float* b, int64_t n) {
    double s = 0.0;
    for (int64_t i = 0; i < n; ++i) s += (double)a[i] * (double)b[i];
    return (float)s;
}
static inline float sigmoid_scalar(float x) { return 1.0f / (1.0f + expf(-x)); }  // activations.rs:1-3

extern "C" void orc_lstm(const float* x, int64_t seq_len, int64_t input_size, int64_t hidden, const float* w,
                         const float* r, const float* bias /*[8H] or NULL*/, const float* h0, const float* c0,
                         float* out_y /*[T,H]*/, float* out_h, float* out_c) {
    const int64_t H = hidden, G = 4 * hidden;
    for (int64_t k = 0; k < H; ++k) {
        out_h[k] = h0 ? h0[k] : 0.0f;
        out_c[k] = c0 ? c0[k] : 0.0f;
    }
    float* gates = new float[G];
    for (int64_t t = 0; t < seq_len; ++t) {
        const float* xt = x + t * input_size;
        for (int64_t g = 0; g < G; ++g) {
            float wc = dot64(w + g * input_size, xt, input_size), rc = dot64(r + g * H, out_h, H);
            float bw = bias ? bias[g] : 0.0f, br = bias ? bias[G + g] : 0.0f;
            gates[g] = wc + rc + bw + br;  // rnn.rs:152-154 (left-associated f32 adds)
        }
        if (orc_scalar_mode()) {
            orc_scalar_lstm_gates(gates, H, out_c, out_h, out_y + t * H);
            continue;
        }
        int64_t k = 0;
        for (; k + 8 <= H; k += 8) {  // lstm_gates_avx2, rnn.rs:15-65; gate order i, o, f, c
            __m256 ig = sigmoid_ps(_mm256_loadu_ps(gates + k)), og = sigmoid_ps(_mm256_loadu_ps(gates + H + k));
            __m256 fg = sigmoid_ps(_mm256_loadu_ps(gates + 2 * H + k)), cg = tanh_ps(_mm256_loadu_ps(gates + 3 * H + k));
            __m256 ct = _mm256_fmadd_ps(fg, _mm256_loadu_ps(out_c + k), _mm256_mul_ps(ig, cg));
            __m256 ht = _mm256_mul_ps(og, tanh_ps(ct));
            _mm256_storeu_ps(out_c + k, ct);
            _mm256_storeu_ps(out_h + k, ht);
            _mm256_storeu_ps(out_y + t * H + k, ht);
        }
        for (; k < H; ++k) {
            float ig = sigmoid_scalar(gates[k]), og = sigmoid_scalar(gates[H + k]), fg = sigmoid_scalar(gates[2 * H + k]);
            float cg = tanhf(gates[3 * H + k]);
            float ct = fg * out_c[k] + ig * cg;
            float ht = og * tanhf(ct);
            out_c[k] = ct;
            out_h[k] = ht;
            out_y[t * H + k] = ht;
        }
    }
    delete[] gates;
}

extern "C" void orc_gru(const float* x, int64_t seq_len, int64_t input_size, int64_t hidden, const float* w,
                        const float* r, const float* bias /*[6H] or NULL*/, const float* h0, float* out_y,
                        float* out_h) {
    // x86 always evaluates the linear_before_reset=1 form (rnn.rs:311-316, 393-407), like the reference's own
    // ref_gru_step oracle (tests/regression_kernels.rs:602-633)
    const int64_t H = hidden, G = 3 * hidden;
    for (int64_t k = 0; k < H; ++k) out_h[k] = h0 ? h0[k] : 0.0f;
    float *wc = new float[G], *rc = new float[G], *hn = new float[H];
    const __m256 one = _mm256_set1_ps(1.0f);
    for (int64_t t = 0; t < seq_len; ++t) {
        const float* xt = x + t * input_size;
        for (int64_t g = 0; g < G; ++g) {
            wc[g] = dot64(w + g * input_size, xt, input_size);
            rc[g] = dot64(r + g * H, out_h, H);
        }
        const float* bw = bias;
        const float* br = bias ? bias + G : nullptr;
        auto B = [&](const float* b, int64_t i) { return b ? b[i] : 0.0f; };
        if (orc_scalar_mode()) {
            orc_scalar_gru_gates(wc, rc, bw, br, H, out_h);
            for (int64_t k2 = 0; k2 < H; ++k2) out_y[t * H + k2] = out_h[k2];
            continue;
        }
        int64_t k = 0;
        for (; k + 8 <= H; k += 8) {  // gru_gate_fusion_avx2, rnn.rs:359-432
            float tmp[8];
            auto ld = [&](const float* p, int64_t off, bool isb) {
                for (int e = 0; e < 8; ++e) tmp[e] = isb ? B(p, off + e) : p[off + e];
                return _mm256_loadu_ps(tmp);
            };
            __m256 z = sigmoid_ps(_mm256_add_ps(_mm256_add_ps(ld(wc, k, false), ld(rc, k, false)),
                                                _mm256_add_ps(ld(bw, k, true), ld(br, k, true))));
            __m256 rg = sigmoid_ps(_mm256_add_ps(_mm256_add_ps(ld(wc, H + k, false), ld(rc, H + k, false)),
                                                 _mm256_add_ps(ld(bw, H + k, true), ld(br, H + k, true))));
            __m256 whx = _mm256_add_ps(ld(wc, 2 * H + k, false), ld(bw, 2 * H + k, true));
            __m256 rrh = _mm256_mul_ps(rg, _mm256_add_ps(ld(rc, 2 * H + k, false), ld(br, 2 * H + k, true)));
            __m256 hg = tanh_ps(_mm256_add_ps(whx, rrh));
            __m256 ht = _mm256_fmadd_ps(_mm256_sub_ps(one, z), hg, _mm256_mul_ps(z, _mm256_loadu_ps(out_h + k)));
            _mm256_storeu_ps(hn + k, ht);
        }
        for (; k < H; ++k) {
            float z = sigmoid_scalar(wc[k] + rc[k] + B(bw, k) + B(br, k));
            float rg = sigmoid_scalar(wc[H + k] + rc[H + k] + B(bw, H + k) + B(br, H + k));
            float hg = tanhf((wc[2 * H + k] + B(bw, 2 * H + k)) + rg * (rc[2 * H + k] + B(br, 2 * H + k)));
            hn[k] = (1.0f - z) * hg + z * out_h[k];
        }
        for (int64_t i = 0; i < H; ++i) {
            out_h[i] = hn[i];
            out_y[t * H + i] = hn[i];
        }
    }
    delete[] wc;
    delete[] rc;
    delete[] hn;
}

// Convolutions: ONNX semantics (= the reference's in-test oracle ref_conv2d, tests/regression_kernels.rs:23-69),
// float64-accumulated, then the x86 epilogue per (n, oc) plane: + bias, then ReLU / SiLU with the polynomial in the
// 8-wide body and libm in the tail (bias_{silu,relu,add}_inplace, avx/math.rs:344-529; conv2d.rs:373-396).
extern "C" void orc_conv2d(const float* x, const float* w, const float* bias, int64_t n, int64_t c, int64_t ih,
                           int64_t iw, int64_t oc, int64_t kh, int64_t kw, int64_t group, int64_t pt, int64_t pl,
                           int64_t pb, int64_t pr, int64_t sh, int64_t sw, int64_t dh, int64_t dw, int act, float* out) {
    const int64_t oh = (ih + pt + pb - dh * (kh - 1) - 1) / sh + 1, ow = (iw + pl + pr - dw * (kw - 1) - 1) / sw + 1;
    const int64_t icg = c / group, ocg = oc / group, plane = oh * ow;
    for (int64_t b = 0; b < n; ++b)
        for (int64_t o = 0; o < oc; ++o) {
            const int64_t g = o / ocg;
            float* op = out + (b * oc + o) * plane;
            for (int64_t y = 0; y < oh; ++y)
                for (int64_t xx = 0; xx < ow; ++xx) {
                    double acc = 0.0;
                    for (int64_t ci = 0; ci < icg; ++ci)
                        for (int64_t a = 0; a < kh; ++a) {
                            const int64_t iy = y * sh - pt + a * dh;
                            if (iy < 0 || iy >= ih) continue;
                            for (int64_t bb = 0; bb < kw; ++bb) {
                                const int64_t ix = xx * sw - pl + bb * dw;
                                if (ix < 0 || ix >= iw) continue;
                                acc += (double)x[((b * c + g * icg + ci) * ih + iy) * iw + ix] *
                                       (double)w[((o * icg + ci) * kh + a) * kw + bb];
                            }
                        }
                    float v = (float)acc;
                    if (bias) v = v + bias[o];
                    op[y * ow + xx] = v;
                }
            if (act == 1)
                for (int64_t i = 0; i < plane; ++i) op[i] = op[i] > 0.0f ? op[i] : 0.0f;
            else if (act == 2)
                orc_unary_simd(3, op, op, plane);  // silu: polynomial body over plane&~7, libm tail
        }
}

extern "C" void orc_conv_transpose2d(const float* x, const float* w /*[C,OC,kh,kw]*/, const float* bias, int64_t n,
                                     int64_t c, int64_t ih, int64_t iw, int64_t oc, int64_t kh, int64_t kw, int64_t pt,
                                     int64_t pl, int64_t pb, int64_t pr, int64_t sh, int64_t sw, int64_t dh, int64_t dw,
                                     float* out) {  // conv2d.rs:2952-3128 (group 1, no output_padding)
    const int64_t oh = (ih - 1) * sh - (pt + pb) + dh * (kh - 1) + 1, ow = (iw - 1) * sw - (pl + pr) + dw * (kw - 1) + 1;
    for (int64_t b = 0; b < n; ++b)
        for (int64_t o = 0; o < oc; ++o)
            for (int64_t y = 0; y < oh; ++y)
                for (int64_t xx = 0; xx < ow; ++xx) {
                    double acc = 0.0;
                    for (int64_t a = 0; a < kh; ++a) {
                        const int64_t ty = y + pt - a * dh;
                        if (ty < 0 || ty % sh) continue;
                        const int64_t iy = ty / sh;
                        if (iy >= ih) continue;
                        for (int64_t bb = 0; bb < kw; ++bb) {
                            const int64_t tx = xx + pl - bb * dw;
                            if (tx < 0 || tx % sw) continue;
                            const int64_t ix = tx / sw;
                            if (ix >= iw) continue;
                            for (int64_t ci = 0; ci < c; ++ci)
                                acc += (double)x[((b * c + ci) * ih + iy) * iw + ix] *
                                       (double)w[((ci * oc + o) * kh + a) * kw + bb];
                        }
                    }
                    float v = (float)acc;
                    if (bias) v = v + bias[o];
                    out[((b * oc + o) * oh + y) * ow + xx] = v;
                }
}
