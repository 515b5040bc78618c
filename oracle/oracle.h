/*
 * oracle.h -- CPU restatement of miuda-ai/lele's hot path (x86_64 AVX2+FMA branch).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library, and only as the checker / the reported CPU baseline -- never as the
 * thing measured or shipped.  lele_amd/ never links, imports or dlopens it.
 *
 * Parity status: PINNED against every golden vector / known-answer test the
 * reference's own test-suite holds for the path (tests/test_oracle_golden.py lists
 * them with reference file:line).  The reference itself (Rust 2024, nightly) cannot
 * be compiled in this image (no rustc/cargo), so there is no oracle/_ref build.
 * The f32 GEMM inside lele is the third-party crate faer 0.24 (not vendored under
 * /root/reference); the oracle restates GEMM as a k-ordered f32 accumulation and
 * additionally offers an f64-accumulated variant -- bit-level parity with faer's
 * summation order is UNPINNED.
 *
 * All functions are plain C ABI over host pointers.  Citations are file:line in
 * /root/reference.
 */
#ifndef LELE_ORACLE_H
#define LELE_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- build mode (scalar.cpp): 0 = the x86 AVX2 bodies (default), 1 = the cfg(not(any(x86_64, aarch64, wasm32))) bodies for the
 * operators that have one upstream: softmax, layer_norm, the LSTM / GRU gate stages, the unary activations, one-channel conv1d,
 * dynamic_quantize_linear.  Process-wide; tests and bench.py's configs[0] baseline set it around their calls. */
void orc_set_scalar_mode(int on);
int orc_scalar_mode(void);
void orc_scalar_softmax_lastdim(const float* input, float* output, int64_t outer, int64_t len);         /* norm.rs:193-216 */
void orc_scalar_layer_norm(const float* input, const float* gamma, const float* beta, float* output, int64_t norm_size,
                           int64_t outer_size, float epsilon);                                         /* norm.rs:286-306 */
void orc_scalar_lstm_gates(const float* gates, int64_t hidden, float* out_c, float* out_h, float* out_y_t); /* rnn.rs:207-221 */
void orc_scalar_gru_gates(const float* wc, const float* rc, const float* bw, const float* br, int64_t hidden, float* h);
void orc_scalar_conv1d_single_channel(const float* input, const float* weights, const float* bias, int64_t batch, int64_t input_len,
                                      int64_t out_channels, int64_t kernel, int64_t stride, int64_t output_len, int relu,
                                      float* output);                                                  /* conv1d.rs:1578-1615 */
void orc_scalar_dynamic_quantize_linear(const float* x, int64_t len, float* y, float* scale, float* zp); /* quantization.rs:1751-1796 */

/* ---- src/features + src/kernels/fft.rs ------------------------------------------------ */
void orc_hann_window(int64_t size, float* out);                     /* features/window.rs:2-13   */
void orc_precompute_twiddles(int64_t n, float* tw_re, float* tw_im, /* kernels/fft.rs:136-157    */
                             int64_t* bit_rev);
/* mode 0: rfft_forward_f32 (fft.rs:2-49), 1: precomputed scalar (79-134), 2: AVX2 (172-266) */
void orc_rfft(const float* input, int64_t n, float* out_re, float* out_im, int mode);
float orc_hz_to_mel_htk(float hz);                                  /* features/mel.rs:1-3       */
float orc_mel_to_hz_htk(float mel);                                 /* features/mel.rs:4-6       */
void orc_mel_filterbank(float sample_rate, int64_t n_fft, int64_t n_mels, float f_min,
                        float f_max /* <0 => None */, float* weights /* [n_mels, n_fft/2+1] */);
/* SparseMelBank::new + apply (mel.rs:48-105) */
void orc_sparse_mel_apply(float sample_rate, int64_t n_fft, int64_t n_mels, float f_min, float f_max,
                          const float* power, float* out);
/* SenseVoiceFrontend::compute (features/pipeline.rs:38-193).  Returns number of LFR rows
 * (0 when pcm_len < frame_len == TensorView::empty()).  `mel_out` (may be NULL) receives the
 * intermediate log-mel [num_frames, n_mels]; `out` receives [t_lfr, n_mels*lfr_m]. */
int64_t orc_frontend_shape(int64_t pcm_len, int64_t sample_rate, float frame_length_ms, float frame_shift_ms,
                           int64_t lfr_n, int64_t* num_frames);
int64_t orc_frontend_compute(const float* pcm, int64_t pcm_len, int64_t sample_rate, int64_t n_mels,
                             float frame_length_ms, float frame_shift_ms, int64_t lfr_m, int64_t lfr_n,
                             float* mel_out, float* out);
void orc_lfr(const float* in, int64_t t, int64_t d, int64_t m, int64_t n, float* out); /* lfr.rs:18-54 */
void orc_cmvn(const float* in, int64_t t, int64_t d, float eps, float* out);          /* cmvn.rs:14-66 */
void orc_cmvn_apply_with_stats(const float* in, int64_t t, int64_t d, float eps, const float* mean,
                               const float* std_, float* out);                         /* cmvn.rs:67-92 */
/* ONNX STFT op (kernels/math.rs:2304-2370) and stft_power_spectrum (2372-2439).
 * window==NULL => periodic Hann over win_length.  Returns num_frames. */
int64_t orc_stft(const float* signal, int64_t len, int64_t n_fft, int64_t hop, int64_t win_length,
                 const float* window, float* out /* [frames, n_fft/2+1, 2] */);
int64_t orc_stft_power(const float* signal, int64_t len, int64_t n_fft, int64_t hop, int64_t win_length,
                       const float* window, float* out /* [frames, n_fft/2+1] */);

/* ---- src/kernels/gemm.rs ---------------------------------------------------------------------------- */
/* acc32 = 0: float64-accumulated inner product (the tolerance reference); 1: k-ordered f32 FMA chain */
void orc_matmul(const float* a, const float* b, int64_t batch_a, int64_t batch_b, int64_t m, int64_t k, int64_t n,
                float* out, int acc32);                                              /* gemm.rs:112-222 */
void orc_matmul_fused_add(const float* a, const float* b, const float* bias, int64_t bias_len, int64_t batch_a,
                          int64_t batch_b, int64_t m, int64_t k, int64_t n, float* out, int acc32); /* 223-432 */
void orc_gemm(const float* a, const float* b, const float* c, int64_t c_len, float alpha, float beta, int trans_a,
              int trans_b, int64_t m, int64_t k, int64_t n, float* out, int acc32); /* gemm.rs:433-535 */

/* ---- src/kernels/quantization.rs (+ avx/quantization.rs) --------------------------------------------- */
void orc_dynamic_quantize_linear(const float* x, int64_t len, float* y, float* scale, float* zp);
void orc_fused_quantized_linear(const float* input, int64_t batch, int64_t m, int64_t k, int64_t n,
                                const float* weight, const float* weight_scale, int64_t weight_scale_len,
                                float weight_zero, const float* bias, int relu, float* out);
void orc_mat_mul_integer(const float* a, const float* b, int64_t batch_a, int64_t batch_b, int64_t m, int64_t k,
                         int64_t n, float zp_a, float zp_b, const float* scale, int64_t scale_len, const float* bias,
                         int relu, float* out);

/* ---- src/kernels/avx/math.rs, avx/norm.rs, norm.rs ---------------------------------------------------- */
/* op: 0 exp 1 sigmoid 2 tanh 3 silu 4 erf 5 gelu 6 fast_gelu 7 relu 8 sqrt (8-wide SIMD body + libm scalar tail) */
void orc_unary_simd(int op, const float* in, float* out, int64_t len);
void orc_layer_norm(const float* input, const float* scale, const float* bias, float* output, int64_t norm_size,
                    int64_t outer_size, float epsilon);
void orc_softmax_lastdim(const float* input, float* output, int64_t outer, int64_t len);
void orc_rms_norm(const float* input, const float* weight, float* output, int64_t norm_size, int64_t outer_size,
                  float epsilon);
void orc_batch_norm(const float* src, const float* scale, const float* bias, const float* mean, const float* var,
                    float epsilon, int64_t outer, int64_t c, int64_t inner, float* out);

/* ---- src/kernels/rnn.rs, conv2d.rs, conv1d.rs ----------------------------------------------------------- */
void orc_lstm(const float* x, int64_t seq_len, int64_t input_size, int64_t hidden, const float* w, const float* r,
              const float* bias, const float* h0, const float* c0, float* out_y, float* out_h, float* out_c);
void orc_gru(const float* x, int64_t seq_len, int64_t input_size, int64_t hidden, const float* w, const float* r,
             const float* bias, const float* h0, float* out_y, float* out_h);
/* act: 0 none, 1 relu, 2 silu; conv1d is the H == 1 case */
void orc_conv2d(const float* x, const float* w, const float* bias, int64_t n, int64_t c, int64_t ih, int64_t iw,
                int64_t oc, int64_t kh, int64_t kw, int64_t group, int64_t pt, int64_t pl, int64_t pb, int64_t pr,
                int64_t sh, int64_t sw, int64_t dh, int64_t dw, int act, float* out);
/* the same convolution the way lele's x86 build computes it: im2col + GEMM + bias / activation pass (conv_fast.cpp) */
void orc_conv2d_im2col(const float* x, const float* w, const float* bias, int64_t n, int64_t c, int64_t ih, int64_t iw,
                       int64_t oc, int64_t kh, int64_t kw, int64_t group, int64_t pt, int64_t pl, int64_t pb, int64_t pr,
                       int64_t sh, int64_t sw, int64_t dh, int64_t dw, int act, float* out);
void orc_conv_transpose2d(const float* x, const float* w, const float* bias, int64_t n, int64_t c, int64_t ih,
                          int64_t iw, int64_t oc, int64_t kh, int64_t kw, int64_t pt, int64_t pl, int64_t pb, int64_t pr,
                          int64_t sh, int64_t sw, int64_t dh, int64_t dw, float* out);

/* ---- application-side steps (apps.cpp): tokenizer.rs:37-86, yolo26n-seg image.rs:62-265, silero main.rs:151-228 ---- */
void orc_decode_greedy_ids(const float* logits, int64_t batch, int64_t steps, int64_t vocab, const uint8_t* skip, int64_t skip_len,
                           int32_t* out, int32_t* counts);
void orc_image_preprocess(const uint8_t* rgb, int64_t height, int64_t width, int64_t target, float* out);
int32_t orc_yolo_seg_postprocess(const float* logits, const float* mask_features, int64_t mask_total, int64_t img_width,
                                 int64_t img_height, float threshold, int64_t num_classes, float* dets, uint8_t* mask_img);
int64_t orc_vad_segments(const float* probs, int64_t num_probs, int64_t chunk_size, int64_t padded_len, int64_t audio_len,
                         uint32_t sample_rate, float threshold, float min_silence_ms, float min_speech_ms, float speech_pad_ms,
                         float merge_gap_ms, int64_t* segments, int64_t max_segments);

#ifdef __cplusplus
}
#endif
#endif
