// scalar.cpp -- the SCALAR build mode of the CPU restatement (TEST INFRASTRUCTURE; see oracle.h).
//
// lele's kernels have three bodies: x86 AVX2 (what this oracle restates everywhere else), aarch64 NEON, and -- under
// #[cfg(not(any(target_arch = "x86_64", target_arch = "aarch64", target_arch = "wasm32")))] -- plain Rust loops.  BASELINE configs[0]
// ("Silero VAD ... plumbing, no GPU") names that last mode.  orc_set_scalar_mode(1) switches
// the operators below to a restatement of exactly those branches; everything else (faer GEMMs, the index operators, shapes) is the
// same code on every architecture upstream and stays as it is here.  Results differ from the AVX2 mode in the last bits only: libm
// exp / tanh instead of the polynomial, round-half-away instead of round-half-even in the quantiser's body, one running sum instead
// of eight lanes.  tests/test_oracle_golden.py holds the two modes against each other and the scalar mode against the reference's own
// scalar oracles (ref_gru_step, tests/regression_kernels.rs:602-633).
#include <math.h>
#include <stdint.h>

#include "oracle.h"

static int g_scalar = 0;
extern "C" void orc_set_scalar_mode(int on) { g_scalar = on != 0; }
extern "C" int orc_scalar_mode(void) { return g_scalar; }

static inline float sigmoid_s(float x) { return 1.0f / (1.0f + expf(-x)); }  // kernels/activations.rs:1-3

// softmax over the last axis, src/kernels/norm.rs:193-216
extern "C" void orc_scalar_softmax_lastdim(const float* input, float* output, int64_t outer, int64_t len) {
    for (int64_t i = 0; i < outer; ++i) {
        const float* src = input + i * len;
        float* dst = output + i * len;
        float max_val = -3.40282347e+38f;  // f32::MIN
        for (int64_t j = 0; j < len; ++j) max_val = src[j] > max_val ? src[j] : (max_val != max_val ? src[j] : max_val);  // f32::max: NaN loses
        float sum = 0.0f;
        for (int64_t j = 0; j < len; ++j) {
            const float e = expf(src[j] - max_val);
            dst[j] = e;
            sum += e;
        }
        const float inv_sum = 1.0f / sum;
        for (int64_t j = 0; j < len; ++j) dst[j] *= inv_sum;
    }
}

// layer_norm, src/kernels/norm.rs:286-306: one pass for sum and sum of squares, var = E[x^2] - mean^2
extern "C" void orc_scalar_layer_norm(const float* input, const float* gamma, const float* beta, float* output, int64_t norm_size,
                                      int64_t outer_size, float epsilon) {
    const float inv_n = 1.0f / (float)norm_size;
    for (int64_t i = 0; i < outer_size; ++i) {
        const float* chunk = input + i * norm_size;
        float* out = output + i * norm_size;
        float sum = 0.0f, sumsq = 0.0f;
        for (int64_t j = 0; j < norm_size; ++j) {
            sum += chunk[j];
            sumsq += chunk[j] * chunk[j];
        }
        const float mean = sum * inv_n;
        const float var = sumsq * inv_n - mean * mean;
        const float inv_std = 1.0f / sqrtf(var + epsilon);
        for (int64_t j = 0; j < norm_size; ++j) out[j] = (chunk[j] - mean) * inv_std * gamma[j] + beta[j];
    }
}

// the LSTM's gate stage, src/kernels/rnn.rs:207-221 (gate order i, o, f, c; activations.rs sigmoid / tanh = libm)
extern "C" void orc_scalar_lstm_gates(const float* gates, int64_t hidden, float* out_c, float* out_h, float* out_y_t) {
    for (int64_t k = 0; k < hidden; ++k) {
        const float i_gate = sigmoid_s(gates[k]);
        const float o_gate = sigmoid_s(gates[hidden + k]);
        const float f_gate = sigmoid_s(gates[2 * hidden + k]);
        const float c_gate = tanhf(gates[3 * hidden + k]);
        const float ct = f_gate * out_c[k] + i_gate * c_gate;
        const float ht = o_gate * tanhf(ct);
        out_c[k] = ct;
        out_h[k] = ht;
        out_y_t[k] = ht;
    }
}

// one GRU step.  The scalar branch upstream (rnn.rs:319-349) reads the hidden gate's recurrent term at H + k instead of 2 H + k
// (rnn.rs:330: a defect); the reference's own scalar oracle for the operator is ref_gru_step
// (tests/regression_kernels.rs:602-633, linear_before_reset = true: what x86 evaluates), restated here.  wc / rc: the two GEMV
// results [3 H]; bw / br: the two bias halves or NULL.
extern "C" void orc_scalar_gru_gates(const float* wc, const float* rc, const float* bw, const float* br, int64_t hidden, float* h) {
    auto B = [](const float* b, int64_t i) { return b ? b[i] : 0.0f; };
    for (int64_t k = 0; k < hidden; ++k) {
        const float z = sigmoid_s(wc[k] + rc[k] + B(bw, k) + B(br, k));
        const float r = sigmoid_s(wc[hidden + k] + rc[hidden + k] + B(bw, hidden + k) + B(br, hidden + k));
        const float h_pre = wc[2 * hidden + k] + B(bw, 2 * hidden + k) + r * (rc[2 * hidden + k] + B(br, 2 * hidden + k));
        const float h_gate = tanhf(h_pre);
        h[k] = (1.0f - z) * h_gate + z * h[k];
    }
}

// conv1d of ONE input channel, any kernel / stride / output channels, src/kernels/conv1d.rs:1578-1615: a running sum over the taps,
// then bias, then ReLU (s.max(0.0): NaN becomes 0)
extern "C" void orc_scalar_conv1d_single_channel(const float* input, const float* weights, const float* bias, int64_t batch,
                                                 int64_t input_len, int64_t out_channels, int64_t kernel, int64_t stride,
                                                 int64_t output_len, int relu, float* output) {
    for (int64_t b = 0; b < batch; ++b) {
        const float* in_base = input + b * input_len;
        float* out_base = output + b * out_channels * output_len;
        for (int64_t oc = 0; oc < out_channels; ++oc) {
            const float* w = weights + oc * kernel;
            for (int64_t t = 0; t < output_len; ++t) {
                const float* in = in_base + t * stride;
                float s = 0.0f;
                for (int64_t k = 0; k < kernel; ++k) s += in[k] * w[k];
                if (bias) s += bias[oc];
                if (relu) s = s > 0.0f ? s : 0.0f;
                out_base[oc * output_len + t] = s;
            }
        }
    }
}

// dynamic_quantize_linear, src/kernels/quantization.rs:1751-1796: x * inv_scale + zp as TWO roundings, f32::round (half away from
// zero) for every element -- the AVX2 body rounds fma(x, inv_scale, zp) half to even
extern "C" void orc_scalar_dynamic_quantize_linear(const float* x, int64_t len, float* y, float* scale, float* zp) {
    float min_val = 3.40282347e+38f, max_val = -3.40282347e+38f;
    for (int64_t i = 0; i < len; ++i) {
        if (x[i] < min_val) min_val = x[i];
        if (x[i] > max_val) max_val = x[i];
    }
    const float adjusted_max = max_val > 0.0f ? max_val : 0.0f, adjusted_min = min_val < 0.0f ? min_val : 0.0f;
    float range = adjusted_max - adjusted_min;
    if (!(range > 1e-5f)) range = 1e-5f;
    const float s = range / 255.0f;
    float z = roundf(-adjusted_min / s);
    z = z < 0.0f ? 0.0f : (z > 255.0f ? 255.0f : z);
    const float inv = 1.0f / s;
    *scale = s;
    *zp = z;
    for (int64_t i = 0; i < len; ++i) {
        float q = roundf(x[i] * inv + z);
        y[i] = q < 0.0f ? 0.0f : (q > 255.0f ? 255.0f : q);
    }
}
