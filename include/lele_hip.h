/*
 * lele_hip.h -- C ABI of liblele_hip.so: an MI355X (gfx950) implementation of lele's hot path.
 *
 * Every entry point is what lele's FFI for the path would bind.  lele (Rust) has no GPU back end; the
 * precedent for a C-ABI kernel boundary in the reference is its macOS cblas_sgemm binding
 * (/root/reference/src/kernels/gemm.rs:30-47).  Each function below names the reference function it
 * replaces (file:line under /root/reference).  INTEGRATION.md shows the Rust `extern "C"` stub for each.
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types.
 *   - every function returns 0 on success, non-zero on error; lele_hip_last_error() then describes it.
 *     lele's kernels panic! on shape/attr violations (e.g. rnn.rs:85-90, norm.rs:218); the Rust shim turns
 *     a non-zero code into the same panic, nothing unwinds across the FFI.
 *   - tensors are row-major, described by LeleTensor {data, shape, rank, dtype, mem}: the C image of
 *     lele::tensor::TensorView (src/tensor.rs:5-12).  mem says where `data` lives:
 *       LELE_MEM_HOST   : host memory, staged to the device for the call (drop-in semantics);
 *       LELE_MEM_DEVICE : device memory (e.g. lele_hip_buf_data() of a previous op's output);
 *       LELE_MEM_WEIGHT : host memory that is immutable for the life of the ctx (weights.bin slices):
 *                         uploaded (and pre-packed where an op needs it) once, cached by (ptr, bytes) --
 *                         the analogue of lele's thread-local B_WEIGHT_CACHE (avx/quantization.rs:12-95).
 *   - outputs go to a LeleBuf: a growable device allocation that mirrors the `out: &mut Vec<f32>` workspace
 *     buffers of generated code (src/compiler/mod.rs:148-290).  The op resizes it, writes the result and
 *     reports the result shape through out_shape[0..*out_rank) (capacity LELE_MAX_RANK).
 *   - a LeleCtx owns one HIP stream; all work of a ctx is stream-ordered; lele_hip_sync() drains it.
 *     One ctx per host thread (lele itself is single-threaded with thread-local scratch, conv2d.rs:601-603).
 */
#ifndef LELE_HIP_H
#define LELE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LELE_MAX_RANK 8

typedef enum { LELE_F32 = 0, LELE_I64 = 1, LELE_I32 = 2, LELE_U8 = 3, LELE_I8 = 4 } LeleDType;
typedef enum { LELE_MEM_HOST = 0, LELE_MEM_DEVICE = 1, LELE_MEM_WEIGHT = 2 } LeleMem;
typedef enum { LELE_ACT_NONE = 0, LELE_ACT_RELU = 1, LELE_ACT_SILU = 2 } LeleAct;

typedef struct LeleTensor {
    const void* data;
    const int64_t* shape;
    int32_t rank;
    int32_t dtype; /* LeleDType */
    int32_t mem;   /* LeleMem   */
} LeleTensor;

/* Channel views.  A Concat along C of NCHW tensors and a Split along C move no value: an operand of the Concat is the window
 * [c0, c1) of the result, a result of the Split the window [c0, c1) of its operand -- image n of such a window starts
 * n * pitch elements after image 0 (pitch = C_total * H * W of the enclosing tensor) and is dense inside.  The *_pitched entry
 * points take operands and / or write results of that form, so that the convolution / addition / pooling / resize that produces a
 * Concat operand writes it in place and the one that consumes a Split result reads it in place (lele copies: manipulation.rs:108-207,
 * 1091-1151; the values are the same).  All fields in ELEMENTS; 0 = dense.
 *   x_pitch, y_pitch : first / second tensor operand (its `data` already points at the window's first element)
 *   out_offset, out_pitch : the result goes to out's data + out_offset, images out_pitch apart; `out` must ALREADY hold the
 *                       enclosing tensor (lele_hip_buf_reserve) -- it is not resized, its other contents are untouched.  With
 *                       out_pitch == 0 the op resizes `out` and writes a dense result, as its plain form does. */
typedef struct LelePitch {
    int64_t x_pitch;
    int64_t y_pitch;
    int64_t out_offset;
    int64_t out_pitch;
} LelePitch;

typedef struct LeleCtx LeleCtx;
typedef struct LeleBuf LeleBuf;
typedef struct LeleFrontend LeleFrontend;
typedef struct LeleGraph LeleGraph;

/* ---- context / memory ------------------------------------------------------------------------------ */
const char* lele_hip_last_error(void);
int lele_hip_device_count(int* count);
int lele_hip_ctx_create(int device, LeleCtx** out);
int lele_hip_ctx_destroy(LeleCtx* ctx);
int lele_hip_sync(LeleCtx* ctx);
void* lele_hip_ctx_stream(LeleCtx* ctx); /* hipStream_t */
/* hipGraph capture of an op sequence.  lele's generated forward() is a fixed sequence of kernel calls per input shape
 * (src/compiler/mod.rs:1291-1303); at SenseVoice's token counts every call is launch-latency bound, so the sequence is
 * recorded once and replayed as ONE graph launch.  Between begin and end every lele_hip_* op on this ctx is recorded
 * instead of executed; ops must not allocate, grow a LeleBuf, synchronise, or take LELE_MEM_HOST inputs while
 * capturing (they return an error) -- run the sequence once eagerly first, with the same buffers and shapes.  Replays
 * read and write the same device buffers as the recorded calls. */
int lele_hip_graph_begin(LeleCtx* ctx);
int lele_hip_graph_end(LeleCtx* ctx, LeleGraph** out);
int lele_hip_graph_abort(LeleCtx* ctx);
int lele_hip_graph_launch(LeleGraph* graph);
int lele_hip_graph_destroy(LeleGraph* graph);
/* Lanes: a ctx owns up to 4 HIP streams.  lele's generated forward() is a SEQUENCE, but the graph behind it is not: the FSMN memory
 * block of a SenseVoice layer does not depend on the attention beside it, the three detection heads of Yolo26n-seg not on each other
 * (examples/yolo26n-seg/src/yolo26seg.rs:509-627).  A plan runner that knows the dependencies issues independent branches on
 * different lanes; recorded between graph_begin and graph_end they become parallel branches of ONE hipGraph (while capturing, a lane
 * is a chain of graph nodes and an event the set of nodes it stands for: node-to-node edges on the one capturing stream).
 *   lane_set    : ops issued from now on go to `lane` (0 = the ctx stream of lele_hip_ctx_stream).  A lane has its own stream,
 *                 staging arena, scratch block and temporary buffers; it is created on first use (not while capturing: run the
 *                 sequence once eagerly with its lanes, as for every other allocation).  Results are ordered between lanes ONLY by
 *                 record / wait: the caller orders every cross-lane read-after-write, write-after-read and write-after-write.
 *   lane_record : event := everything issued on the current lane so far.
 *   lane_wait   : the current lane continues only after that point.
 * lele_hip_sync drains every lane; buf_to_host / graph_launch / the communicator use the CURRENT lane (call them on lane 0). */
int lele_hip_lane_set(LeleCtx* ctx, int lane);
int lele_hip_lane_record(LeleCtx* ctx, int event);
int lele_hip_lane_wait(LeleCtx* ctx, int event);
/* stream-ordered stopwatch (HIP events on the ctx stream) used by bench.py */
int lele_hip_timer_start(LeleCtx* ctx);
int lele_hip_timer_stop(LeleCtx* ctx, float* elapsed_ms);

int lele_hip_buf_create(LeleCtx* ctx, LeleBuf** out);
int lele_hip_buf_destroy(LeleBuf* buf);
int lele_hip_buf_reserve(LeleBuf* buf, size_t bytes);
void* lele_hip_buf_data(LeleBuf* buf);
/* Tell the library that the buffer's contents were written behind its back (through lele_hip_buf_data() + the ctx stream: a
 * hipMemcpy, an RCCL receive, the integrator's own kernel).  Producer-side statistics kept next to the buffer (the {min, max}
 * row pairs a LayerNorm / linear leaves for the dynamic quantisation that reads the tensor next) are dropped.  Every op of this
 * library does the equivalent on its own outputs. */
int lele_hip_buf_mark_dirty(LeleBuf* buf);
size_t lele_hip_buf_bytes(LeleBuf* buf); /* size of the last result in bytes */
int lele_hip_buf_from_host(LeleBuf* buf, const void* src, size_t bytes);
int lele_hip_buf_to_host(LeleBuf* buf, void* dst, size_t bytes); /* synchronises the ctx stream */

/* ---- src/features ---------------------------------------------------------------------------------- */
/* FeatureConfig, src/features/pipeline.rs:8-27 */
typedef struct LeleFeatureConfig {
    int64_t sample_rate;
    int64_t n_mels;
    float frame_length_ms;
    float frame_shift_ms;
    int64_t lfr_m;
    int64_t lfr_n;
} LeleFeatureConfig;

/* SenseVoiceFrontend::new, pipeline.rs:38-65 (window, twiddle/bit-reverse tables, sparse mel bank, LFR) */
/* hann_window (window.rs:2-13) and mel_filterbank (mel.rs:7-56; f_max = sample_rate / 2 unless has_f_max) as the library builds
 * them for its own front-end: host arithmetic into host memory (`size` floats / n_mels * (n_fft / 2 + 1) floats), no ctx. */
int lele_hip_hann_window(int64_t size, float* out);
float lele_hip_hz_to_mel_htk(float hz);   /* mel.rs:1-3 */
float lele_hip_mel_to_hz_htk(float mel);  /* mel.rs:4-6 */
int lele_hip_mel_filterbank(float sample_rate, int64_t n_fft, int64_t n_mels, float f_min, int32_t has_f_max, float f_max, float* out);
int lele_hip_frontend_create(LeleCtx* ctx, const LeleFeatureConfig* cfg, LeleFrontend** out);
int lele_hip_frontend_destroy(LeleFrontend* fe);
/* rows of the [T, n_mels*lfr_m] result for a pcm of `pcm_len` samples; 0 <=> TensorView::empty() (pipeline.rs:70-73) */
int lele_hip_frontend_out_rows(const LeleFrontend* fe, int64_t pcm_len, int64_t* rows, int64_t* cols,
                               int64_t* num_frames);
/* SenseVoiceFrontend::compute, pipeline.rs:67-193.  pcm: f32 [pcm_len] -> out [T, n_mels*lfr_m] */
int lele_hip_frontend_compute(LeleFrontend* fe, const LeleTensor* pcm, LeleBuf* out, int64_t* out_shape,
                              int32_t* out_rank);
/* The same for `batch` equal-length utterances stored back to back: pcm [batch, pcm_len] ->
 * out [batch, T, n_mels*lfr_m].  (lele loops over utterances on the host; examples/sensevoice/src/main.rs:58-86) */
int lele_hip_frontend_compute_batch(LeleFrontend* fe, const LeleTensor* pcm, LeleBuf* out, int64_t* out_shape,
                                    int32_t* out_rank);
/* log-mel before LFR ([num_frames, n_mels]) of the last compute call's shape, for tests */
int lele_hip_frontend_logmel(LeleFrontend* fe, const LeleTensor* pcm, LeleBuf* out, int64_t* out_shape,
                             int32_t* out_rank);
/* per-kernel stopwatch for bench.py's roofline block: while on, every compute call records HIP events around
 * its two kernels on the ctx stream (no synchronisation); profile_read() drains the stream and returns the
 * average duration of fe_frame_sum_kernel and fe_main_kernel over the `runs` calls since the last read. */
int lele_hip_frontend_set_profiling(LeleFrontend* fe, int on);
int lele_hip_frontend_profile_read(LeleFrontend* fe, float* sum_kernel_ms, float* main_kernel_ms, int64_t* runs);

/* Lfr::compute, src/features/lfr.rs:18-54: [T, D] (or [1,T,D]) -> [ceil(T/n), D*m] */
int lele_hip_lfr(LeleCtx* ctx, const LeleTensor* x, int64_t m, int64_t n, LeleBuf* out, int64_t* out_shape,
                 int32_t* out_rank);
/* Cmvn::compute, src/features/cmvn.rs:14-66: per-utterance mean/var over time.  [T,D] or [1,T,D] as upstream; extension:
 * [B,T,D] with B > 1 normalises each of the B utterances with its own statistics in one launch pair */
int lele_hip_cmvn(LeleCtx* ctx, const LeleTensor* x, float eps, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* Cmvn::apply_with_stats, cmvn.rs:67-92 */
int lele_hip_cmvn_apply_with_stats(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* mean, const LeleTensor* std_,
                                   float eps, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* RealFft::process (features/fft.rs:18-43) / rfft_forward_f32_precomputed (kernels/fft.rs:51-77):
 * x [rows, n] real, n power of two -> re, im [rows, n/2+1] */
int lele_hip_rfft(LeleCtx* ctx, const LeleTensor* x, LeleBuf* out_re, LeleBuf* out_im, int64_t* out_shape,
                  int32_t* out_rank);
/* kernels::stft / stft_power_spectrum, src/kernels/math.rs:2304-2439; window may be NULL (periodic Hann) */
int lele_hip_stft(LeleCtx* ctx, const LeleTensor* signal, int64_t n_fft, int64_t hop_length, int64_t win_length,
                  const LeleTensor* window, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
int lele_hip_stft_power_spectrum(LeleCtx* ctx, const LeleTensor* signal, int64_t n_fft, int64_t hop_length,
                                 int64_t win_length, const LeleTensor* window, LeleBuf* out, int64_t* out_shape,
                                 int32_t* out_rank);

/* ---- src/kernels/gemm.rs ---------------------------------------------------------------------------- */
/* matmul, gemm.rs:112-222: [..,M,K] x [..,K,N]; B may be un-batched (batch_b == 1) else batches must match */
int lele_hip_matmul(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, LeleBuf* out, int64_t* out_shape,
                    int32_t* out_rank);
/* matmul_fused_add, gemm.rs:223-432: bias.len()==N -> per-column bias, else out[i] += bias[i % len] */
int lele_hip_matmul_fused_add(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, const LeleTensor* bias,
                              LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* gemm, gemm.rs:433-535: 2-D, out[M,N] = alpha*op(A)*op(B) + beta*C (C broadcast: full / [N] / [M] / scalar / modulo) */
int lele_hip_gemm(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, const LeleTensor* c_or_null, float alpha,
                  float beta, int trans_a, int trans_b, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* ---- src/kernels/quantization.rs --------------------------------------------------------------------- */
/* fused_quantized_linear, quantization.rs:77-169: DynamicQuantizeLinear (one range PER BATCH SLICE) + MatMulInteger
 * + scale + bias [+ ReLU].  weight_int8: u8 values carried as f32 [K,N]; weight_scale [1] or [N]; weight_zero [1];
 * bias [N] or empty/NULL.  Bit-exact with the reference's x86 path. */
int lele_hip_fused_quantized_linear(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8,
                                    const LeleTensor* weight_scale, const LeleTensor* weight_zero,
                                    const LeleTensor* bias_or_null, int apply_relu, LeleBuf* out, int64_t* out_shape,
                                    int32_t* out_rank);
/* dynamic_quantize_linear, quantization.rs:1628-1657: y (u8 values as f32, shape of x), scale [1], zero_point [1] */
int lele_hip_dynamic_quantize_linear(LeleCtx* ctx, const LeleTensor* x, LeleBuf* out_y, LeleBuf* out_scale,
                                     LeleBuf* out_zp, int64_t* out_shape, int32_t* out_rank);
/* mat_mul_integer / _with_bias / _with_scale_bias / _with_scale_bias_relu, quantization.rs:8-72, 927-992:
 * a, b hold u8 values as f32; zero points [1], scale [1] or [N], bias [N] -- each may be NULL */
int lele_hip_mat_mul_integer_with_scale_bias(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b,
                                             const LeleTensor* a_zero_point, const LeleTensor* b_zero_point,
                                             const LeleTensor* scale, const LeleTensor* bias, int apply_relu,
                                             LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* prepare_weights (quantization.rs:221) / PreparedWeights: a weight matrix given as RAW u8 bytes [K, N] (row-major), packed for
 * the i8 matrix cores once.  mat_mul_integer_prepared (quantization.rs:699): `a` holds u8 values as f32; zero points are host
 * scalars (has_* == 0 <=> None).  fused_dq_gemm_prepared (fused_dq_gemm_prepared_x86, quantization.rs:454): dynamic
 * quantisation per batch slice + prepared GEMM + scale + bias [+ ReLU] -- fused_quantized_linear on a prepared matrix.
 * mat_mul_integer_u8_weights (quantization.rs:173) is prepare_weights + mat_mul_integer_prepared; the shim keeps the handle. */
typedef struct LelePrepared LelePrepared;
int lele_hip_prepare_weights(LeleCtx* ctx, const uint8_t* b_u8, int64_t k, int64_t n, LelePrepared** out);
int lele_hip_prepared_destroy(LelePrepared* pw);
int lele_hip_mat_mul_integer_prepared(LeleCtx* ctx, const LeleTensor* a, const LelePrepared* pw, int has_a_zero_point,
                                      float a_zero_point, int has_b_zero_point, int32_t b_zero_point, const LeleTensor* scale,
                                      const LeleTensor* bias, int apply_relu, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
int lele_hip_fused_dq_gemm_prepared(LeleCtx* ctx, const LeleTensor* input, const LelePrepared* pw, int has_b_zero_point,
                                    int32_t b_zero_point, const LeleTensor* weight_scale, const LeleTensor* bias, int apply_relu,
                                    LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* per-stage stopwatch of fused_quantized_linear for bench.py's roofline block: while on, every EAGER call records HIP events on
 * the ctx stream between its stages (range pass | row quantisation | i8 GEMM; a stage a call skips reads 0); profile_read()
 * drains the stream and returns the average duration of each stage over the calls since the last read. */
int lele_hip_quant_set_profiling(LeleCtx* ctx, int on);
int lele_hip_quant_profile_read(LeleCtx* ctx, float* range_ms, float* quantise_ms, float* gemm_ms, int64_t* calls);

/* ---- src/kernels/math.rs: activations and element-wise ops ---------------------------------------------- */
/* Unary f32 ops.  LeleUnaryOp names the lele kernel it replaces (math.rs file:line in eltwise.hip). */
typedef enum {
    LELE_U_EXP = 0, LELE_U_SIGMOID = 1, LELE_U_TANH = 2, LELE_U_SILU = 3, LELE_U_ERF = 4, LELE_U_GELU = 5,
    LELE_U_FAST_GELU = 6, LELE_U_RELU = 7, LELE_U_SQRT = 8, LELE_U_LOG = 9, LELE_U_SIN = 10, LELE_U_COS = 11,
    LELE_U_NEG = 12, LELE_U_RECIPROCAL = 13, LELE_U_SOFTPLUS = 14, LELE_U_NOT = 15, LELE_U_ABS = 16,
    LELE_U_FLOOR = 17, LELE_U_CEIL = 18
} LeleUnaryOp;
int lele_hip_unary(LeleCtx* ctx, int op, const LeleTensor* x, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* Binary ops with numpy-style broadcasting (math.rs:69-264): add, sub, mul, div work on f32 and i64 */
typedef enum {
    LELE_B_ADD = 0, LELE_B_SUB = 1, LELE_B_MUL = 2, LELE_B_DIV = 3, LELE_B_POW = 4, LELE_B_MAX = 5, LELE_B_MIN = 6,
    LELE_B_EQUAL = 7, LELE_B_LESS = 8, LELE_B_GREATER = 9, LELE_B_PRELU = 10, LELE_B_MOD = 11, LELE_B_AND = 12,
    LELE_B_OR = 13
} LeleBinaryOp;
int lele_hip_binary(LeleCtx* ctx, int op, const LeleTensor* a, const LeleTensor* b, LeleBuf* out, int64_t* out_shape,
                    int32_t* out_rank);
/* the same op on SAME-SHAPE f32 operands that are channel views and / or with a channel-view result (LelePitch above: a residual
 * add whose operand is a Split result and whose result is a Concat operand; math.rs:414 on copies) */
int lele_hip_binary_pitched(LeleCtx* ctx, int op, const LeleTensor* a, const LeleTensor* b, const LelePitch* pitch, LeleBuf* out,
                            int64_t* out_shape, int32_t* out_rank);
/* where_op, manipulation.rs:1215: out = cond != 0 ? x : y */
int lele_hip_where(LeleCtx* ctx, const LeleTensor* cond, const LeleTensor* x, const LeleTensor* y, LeleBuf* out,
                   int64_t* out_shape, int32_t* out_rank);
/* clip, math.rs:1984-2010 */
int lele_hip_clip(LeleCtx* ctx, const LeleTensor* x, int has_min, float min_v, int has_max, float max_v, LeleBuf* out,
                  int64_t* out_shape, int32_t* out_rank);
/* reduce_sum / reduce_mean / reduce_max / reduce_l2 (/ min), math.rs:1527-1920 */
typedef enum { LELE_R_SUM = 0, LELE_R_MEAN = 1, LELE_R_MAX = 2, LELE_R_L2 = 3, LELE_R_MIN = 4 } LeleReduceOp;
int lele_hip_reduce(LeleCtx* ctx, int op, const LeleTensor* x, const int64_t* axes, size_t naxes, int keepdims,
                    LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* ---- src/kernels/norm.rs --------------------------------------------------------------------------------- */
int lele_hip_layer_norm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, const LeleTensor* bias,
                        int32_t axis, float epsilon, LeleBuf* out, int64_t* out_shape, int32_t* out_rank); /* norm.rs:226 */
int lele_hip_rms_norm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* weight, int32_t axis, float epsilon,
                      LeleBuf* out, int64_t* out_shape, int32_t* out_rank);                                /* norm.rs:420 */
int lele_hip_softmax(LeleCtx* ctx, const LeleTensor* x, int32_t axis, LeleBuf* out, int64_t* out_shape,
                     int32_t* out_rank);                                                                    /* norm.rs:8 */
int lele_hip_batch_norm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, const LeleTensor* bias,
                        const LeleTensor* mean, const LeleTensor* var, float epsilon, LeleBuf* out, int64_t* out_shape,
                        int32_t* out_rank);                                                                  /* norm.rs:313 */

/* ---- src/kernels/manipulation.rs, shape.rs, conv2d.rs:1051-1502: data movement (bit-exact) ---------------- */
/* out[c] = x[offset + sum_k c_k*strides[k]] (c_k taken modulo mods[k] when mods && mods[k] > 0): the engine behind
 * slice (manipulation.rs:209), transpose (644), expand (math.rs:2168), tile (math.rs:2249), split (1091) */
int lele_hip_strided_copy(LeleCtx* ctx, const LeleTensor* x, const int64_t* out_dims, const int64_t* strides,
                          const int64_t* mods_or_null, int32_t rank, int64_t offset, LeleBuf* out, int64_t* out_shape,
                          int32_t* out_rank);
int lele_hip_concat(LeleCtx* ctx, const LeleTensor* const* inputs, size_t ninputs, int64_t axis, LeleBuf* out,
                    int64_t* out_shape, int32_t* out_rank);                                   /* manipulation.rs:108 */
/* pads = [begin_0..begin_{r-1}, end_0..end_{r-1}]; mode 0 constant, 1 edge, 2 reflect; fill_bits = raw element */
int lele_hip_pad(LeleCtx* ctx, const LeleTensor* x, const int64_t* pads, int32_t mode, uint64_t fill_bits,
                 LeleBuf* out, int64_t* out_shape, int32_t* out_rank);                        /* manipulation.rs:382 */
int lele_hip_gather(LeleCtx* ctx, const LeleTensor* data, const LeleTensor* indices, int64_t axis, LeleBuf* out,
                    int64_t* out_shape, int32_t* out_rank);                                   /* manipulation.rs:589 */
int lele_hip_gather_elements(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* indices, int64_t axis, LeleBuf* out,
                             int64_t* out_shape, int32_t* out_rank);                          /* conv2d.rs:1438 */
int lele_hip_resize_nearest(LeleCtx* ctx, const LeleTensor* x, int64_t out_h, int64_t out_w, int asymmetric,
                            LeleBuf* out, int64_t* out_shape, int32_t* out_rank);             /* conv2d.rs:1261 */
int lele_hip_max_pool2d(LeleCtx* ctx, const LeleTensor* x, const int64_t* kernel_shape, size_t nk,
                        const int64_t* strides, size_t ns, const int64_t* pads, size_t np, const int64_t* dilations,
                        size_t nd, int ceil_mode, LeleBuf* out, int64_t* out_shape, int32_t* out_rank); /* conv2d.rs:1051 */
/* resize_nearest / max_pool2d reading and / or writing channel views, and the plain copy between views (LelePitch above):
 * what concat (manipulation.rs:108) and split (manipulation.rs:1091) along C are when an operand cannot be produced in place */
int lele_hip_resize_nearest_pitched(LeleCtx* ctx, const LeleTensor* x, int64_t out_h, int64_t out_w, int asymmetric,
                                    const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
int lele_hip_max_pool2d_pitched(LeleCtx* ctx, const LeleTensor* x, const int64_t* kernel_shape, size_t nk, const int64_t* strides,
                                size_t ns, const int64_t* pads, size_t np, const int64_t* dilations, size_t nd, int ceil_mode,
                                const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
int lele_hip_copy_pitched(LeleCtx* ctx, const LeleTensor* x, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* The transposing copy of a channel view: x [N, C, ...] (image pitch x_pitch) -> [N, P, C] (P = the product of the trailing
 * dimensions), dense or into the window (out_offset, out_pitch) of an already reserved buffer -- rows [p0, p0 + P) of a wider
 * [N, P_total, C] tensor are the window p0 * C, P_total * C.  lele's detection tails are Concat(levels, axis = 2) -> Transpose(0, 2, 1)
 * -> Split(heads, axis = 2) (examples/yolo26n-seg/src/yolo26seg.rs): three passes over the predictions, each a copy; the same
 * values arrive with one of these calls per (level, head) (lele_amd/plan.py, fold_transposed_splits). */
int lele_hip_transpose_cp_pitched(LeleCtx* ctx, const LeleTensor* x, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* adaptive_avg_pool1d, pooling.rs:1-30: x [.., L] -> [.., output_len]; window i = [floor(i*L/O), ceil((i+1)*L/O)) */
int lele_hip_adaptive_avg_pool1d(LeleCtx* ctx, const LeleTensor* x, int64_t output_len, LeleBuf* out, int64_t* out_shape,
                                 int32_t* out_rank);
int lele_hip_topk(LeleCtx* ctx, const LeleTensor* x, int64_t k, int largest, LeleBuf* out_values, LeleBuf* out_indices,
                  int64_t* out_shape, int32_t* out_rank);                                     /* conv2d.rs:1385 */
int lele_hip_range_f32(LeleCtx* ctx, float start, float delta, int64_t n, LeleBuf* out, int64_t* out_shape,
                       int32_t* out_rank);                                                    /* math.rs:2033 */
int lele_hip_range_i64(LeleCtx* ctx, int64_t start, int64_t delta, int64_t n, LeleBuf* out, int64_t* out_shape,
                       int32_t* out_rank);                                                    /* math.rs:2057 */
int lele_hip_fill(LeleCtx* ctx, const int64_t* shape, int32_t rank, int32_t dtype, uint64_t bits, LeleBuf* out,
                  int64_t* out_shape, int32_t* out_rank);                                     /* shape.rs:122 */
int lele_hip_cast(LeleCtx* ctx, const LeleTensor* x, int32_t to_dtype, LeleBuf* out, int64_t* out_shape,
                  int32_t* out_rank);                                                         /* utils.rs:66-101 */

/* ---- src/kernels/conv2d.rs, conv1d.rs: convolutions (implicit GEMM on the f32 MFMA core) ------------------- */
/* conv2d (conv2d.rs:107), conv2d_fused (conv2d.rs:420: act = LELE_ACT_RELU), conv2d_silu (conv2d.rs:437:
 * act = LELE_ACT_SILU).  x [N,C,H,W], w [C_out,C_in/g,kH,kW], bias [C_out] or NULL.  dilations / strides hold 0, 1 or 2
 * values (one value is used for both axes), pads holds 0, 2 ([ph, pw]) or 4 ([top, left, bottom, right]) values.
 * The SiLU of the epilogue is x * rcp(1 + exp2(-x log2 e)) on the transcendental unit: within 1e-5 relative + 1e-7 of the reference's
 * (avx/math.rs polynomial body, libm tail) -- the sum under it is pinned to 1e-4 only.  LELE_HIP_CONV_SILU_EXACT=1 (read per call)
 * selects the replica of the reference's form instead. */
int lele_hip_conv2d(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                    const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                    const int64_t* strides, size_t nstr, int act, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* conv2d on channel views (LelePitch above; group == 1): x may be a Split result read in place, the result a Concat operand
 * written in place.  Same kernels, same arithmetic order as lele_hip_conv2d on dense copies: identical bits. */
int lele_hip_conv2d_pitched(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                            const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                            const int64_t* strides, size_t nstr, int act, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape,
                            int32_t* out_rank);
/* conv2d followed by the Add of a residual block, in one call: out = act(conv(x) + bias) + res, res [N,C_out,H_out,W_out] f32.
 * lele's generated code issues conv2d_silu and then `add` (examples/yolo26n-seg/src/yolo26seg.rs: every bottleneck's
 * `x + cv2(cv1(x))`); the sum is formed from the same two f32 values, so the bits are those of the two calls.  `pitch` may be NULL;
 * with it, x / the result may be channel views as for lele_hip_conv2d_pitched and y_pitch is the residual's image pitch (0 = dense). */
int lele_hip_conv2d_res(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias, const LeleTensor* res,
                        const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                        const int64_t* strides, size_t nstr, int act, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape,
                        int32_t* out_rank);
/* reset_conv_stats / print_conv_stats (conv2d.rs:75, 101 -- no-ops upstream; examples/yolo26n-seg/src/main.rs:64,74 calls them):
 * 2-D convolutions issued on ctx since the last reset, as call count and multiply-accumulate count */
int lele_hip_conv_stats_reset(LeleCtx* ctx);
int lele_hip_conv_stats(LeleCtx* ctx, int64_t* calls, int64_t* macs);
/* conv1d (conv1d.rs:837) / conv1d_fused (conv1d.rs:1464: relu != 0).  x [N,C,L], w [C_out,C_in/g,K], pads [left,right] */
int lele_hip_conv1d(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                    const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                    const int64_t* strides, size_t nstr, int relu, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* conv_transpose (conv2d.rs:2952): x [N,C,H,W], w [C_in,C_out,kH,kW], group must be 1 (error otherwise, as upstream) */
int lele_hip_conv_transpose(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                            const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                            const int64_t* strides, size_t nstr, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* ---- src/kernels/rnn.rs: LSTM / GRU (batch 1, one direction; anything else is an error, as upstream panics) ---- */
/* lstm (rnn.rs:67): x [T,1,I], w [1,4H,I], r [1,4H,H], bias [1,8H] or NULL; gate order i,o,f,c.
 * out_y [T,1,1,H] (shape written to y_shape), out_h / out_c hold [1,1,H]. sequence_lens is ignored (as upstream). */
int lele_hip_lstm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* r, const LeleTensor* bias,
                  const LeleTensor* sequence_lens, const LeleTensor* initial_h, const LeleTensor* initial_c,
                  LeleBuf* out_y, LeleBuf* out_h, LeleBuf* out_c, int64_t* y_shape, int32_t* y_rank);
/* gru (rnn.rs:246): w [1,3H,I], r [1,3H,H], bias [1,6H] or NULL; gate order z,r,h */
int lele_hip_gru(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* r, const LeleTensor* bias,
                 const LeleTensor* initial_h, int linear_before_reset, LeleBuf* out_y, LeleBuf* out_h, int64_t* y_shape,
                 int32_t* y_rank);

/* ---- ConvInteger family, src/kernels/conv2d.rs:1507-2761 ------------------------------------- */
/* conv_integer (conv2d.rs:2216): x, w hold u8 values as f32; zero points are host scalars ([1] or NULL = 0);
 * out = f32 conv2d(x - x_zp, w - w_zp) with zero padding, as the x86 path computes it */
int lele_hip_conv_integer(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* x_zero_point,
                          const LeleTensor* w_zero_point, const int64_t* dilations, size_t ndil, int64_t group,
                          const int64_t* pads, size_t npads, const int64_t* strides, size_t nstr, LeleBuf* out,
                          int64_t* out_shape, int32_t* out_rank);
/* conv_integer_from_f32 (conv2d.rs:2246; nsrc == 1) and conv_integer_from_f32_multi (conv2d.rs:2420; nsrc > 1, sources
 * concatenated along C, upstream always 1x1/s1/p0): DynamicQuantizeLinear over ALL sources, then conv_integer.
 * out_scale receives the quantisation scale ([1] f32, on the device -- upstream returns it as a host float) */
int lele_hip_conv_integer_from_f32(LeleCtx* ctx, const LeleTensor* const* sources, size_t nsrc, const LeleTensor* w,
                                   const LeleTensor* w_zero_point, const int64_t* dilations, size_t ndil, int64_t group,
                                   const int64_t* pads, size_t npads, const int64_t* strides, size_t nstr, LeleBuf* out,
                                   LeleBuf* out_scale, int64_t* out_shape, int32_t* out_rank);
/* fused_scale_bias / fused_scale_bias_silu (conv2d.rs:2636-2761): out = data * scale + bias[c] (then x / (1 + exp(-x)));
 * scale = scale_dev[0] * scale_mul (scale_dev may be NULL: scale = scale_mul) so that the scale produced by
 * conv_integer_from_f32 never has to visit the host.  Passing the buffer of `data` as `out` is the in-place form. */
int lele_hip_fused_scale_bias(LeleCtx* ctx, const LeleTensor* data, const LeleTensor* scale_dev_or_null, float scale_mul,
                              const LeleTensor* bias, int silu, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* ---- application-side pre/post-processing ----------------------------------------------------- */
/* examples/sensevoice/src/audio.rs:52-73: WAV payload bytes -> f32 mono.  16-bit: i16::from_le_bytes / 32768.0;
 * 8-bit: (b - 128) / 128; stereo: (l + r) / 2.  bytes: U8 [n].  out: f32 [frames] */
int lele_hip_wav_to_f32(LeleCtx* ctx, const LeleTensor* bytes, int32_t bits_per_sample, int32_t num_channels, LeleBuf* out,
                        int64_t* out_shape, int32_t* out_rank);
/* examples/sensevoice/src/tokenizer.rs:50-61: arg-max over the last axis; the LAST of equal maxima wins (Iterator::max_by).
 * x: f32 [.., V] -> i32 ids [..]; only the ids need to leave the device */
int lele_hip_argmax_last(LeleCtx* ctx, const LeleTensor* x, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* tokenizer.rs:63-71: the ids the greedy decoder keeps, in frame order -- an id is dropped when it is outside the vocabulary
 * or its skip flag is set (the host sets skip[0] and skip[id] for every "<|...|>" token once per vocabulary).  No CTC
 * collapsing (upstream does none).  ids: i32 [.., T]; skip: U8 [V]; out_ids: i32 [.., T] (kept ids first, then -1);
 * out_counts: i32 [..] */
int lele_hip_token_filter(LeleCtx* ctx, const LeleTensor* ids, const LeleTensor* skip, LeleBuf* out_ids, LeleBuf* out_counts,
                          int64_t* out_shape, int32_t* out_rank);
/* examples/yolo26n-seg/src/image.rs:62-111 (Image::preprocess): PIL-style nearest resize to target x target
 * (src = min(floor((dst + 0.5) * src_size / target), src_size - 1), f32 arithmetic), HWC u8 -> NCHW f32 / 255.
 * rgb: U8 [H, W, 3] -> out f32 [1, 3, target, target] */
int lele_hip_image_preprocess(LeleCtx* ctx, const LeleTensor* rgb, int32_t target, LeleBuf* out, int64_t* out_shape,
                              int32_t* out_rank);
/* image.rs:127-265 (postprocess_segmentation), per image of a batch.  logits: f32 [N, 300, 38] = box(4, 640-space) + score + class id
 * + 32 mask coefficients; mask_features: f32 [N, 32, Hm, Wm].  out_dets: f32 [N, 300, 38], the first out_count[n] rows of image n are
 * its kept detections in query order (box rescaled and clamped to the image, class id clamped to num_classes - 1), the rows behind
 * them zeros (fixed width: what the ranks of a sharded batch exchange, SURVEY.md 8e "C5");
 * out_count: i32 [N]; out_mask: U8 [N, img_height, img_width] (255 where a detection's mask covers the pixel).
 * The counts stay on the device: the call is graph-capturable. */
int lele_hip_yolo_seg_postprocess(LeleCtx* ctx, const LeleTensor* logits, const LeleTensor* mask_features, int32_t img_width,
                                  int32_t img_height, float threshold, int32_t num_classes, LeleBuf* out_dets,
                                  LeleBuf* out_count, LeleBuf* out_mask);

/* ---- multi-GPU: the one exchange step of the utterance-sharded recogniser ------------------------------ */
/* lele is single-process; its route to several devices is one instance per shard of the utterances (one process per GPU here).
 * Nothing is exchanged while computing; at the end ONE all-gather of the decoded token ids (i32) over RCCL / xGMI gives every
 * rank the transcripts of the whole batch.  RCCL is loaded on first use (dlopen): no link-time dependency.
 * The 128-byte unique id is made by rank 0 (comm_unique_id) and handed to the other ranks by the launcher; comm_init_file does
 * that through a file (rank 0 removes what is there and writes [32-byte job token][id] atomically, the others poll for up to
 * timeout_ms for a file carrying THEIR token).  The token is a digest of the environment variable LELE_JOB_ID (else
 * TORCHELASTIC_RUN_ID), which the launcher sets to a value unique to the launch: a file of an earlier job is never accepted.
 * Without a token the file carries an 8-byte beat that rank 0 keeps incrementing while it waits for the others (and removes the
 * file once all have joined): a reader accepts an id only after it has seen two different beats -- proof that the writer is alive,
 * whatever the ages and clocks involved.  comm_read_id_file is the reader's half on its own (no device, no RCCL). */
typedef struct LeleComm LeleComm;
int lele_hip_comm_unique_id(uint8_t* id128);
int lele_hip_comm_init(LeleCtx* ctx, const uint8_t* id128, int rank, int world, LeleComm** out);
int lele_hip_comm_init_file(LeleCtx* ctx, const char* path, int rank, int world, int timeout_ms, LeleComm** out);
int lele_hip_comm_read_id_file(const char* path, int timeout_ms, uint8_t* id128);
int lele_hip_comm_rank(const LeleComm* comm, int* rank, int* world);
/* send: DEVICE i32 tensor of the same element count on every rank -> out [world, count] in rank order, on the ctx stream
 * (graph-capturable; the result is ordered after everything queued before it, e.g. lele_hip_token_filter) */
int lele_hip_comm_allgather_i32(LeleComm* comm, const LeleTensor* send, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* the same for a DEVICE tensor of any element type (configs[4]: every rank's fixed-width detection rows, f32 [images, 300, 38], and their
 * i32 counts -- the (logits, mask_features) pair of examples/yolo26n-seg/src/yolo26seg.rs:706-716 after lele_hip_yolo_seg_postprocess):
 * out [world, ...send's shape] in rank order, moved as bytes */
int lele_hip_comm_allgather(LeleComm* comm, const LeleTensor* send, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* MAX over ranks of a host scalar, in place (row width of ragged shards, a wall time in ns); synchronises the ctx stream */
int lele_hip_comm_allreduce_max_i64(LeleComm* comm, int64_t* value);
int lele_hip_comm_barrier(LeleComm* comm); /* every rank's ctx stream has drained when any rank returns */
int lele_hip_comm_destroy(LeleComm* comm);

/* ---- fused forms beyond the reference's own patterns (emitted by lele_amd.compiler, each bit-identical to the sequence it
 *      replaces; never required by lele-generated code) ---------------------------------------------------------------- */
/* fused_quantized_linear followed by one or two Adds of same-shape tensors, folded into the GEMM's store:
 * ((linear(x) + res1) + res2), each sum rounded as the separate `add`s would (res2 may be NULL) */
int lele_hip_fused_quantized_linear_residual(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8,
                                             const LeleTensor* weight_scale, const LeleTensor* weight_zero, const LeleTensor* bias,
                                             int apply_relu, const LeleTensor* res1, const LeleTensor* res2, LeleBuf* out,
                                             int64_t* out_shape, int32_t* out_rank);
/* the same followed by the LayerNorm over the last axis that reads the sum (src/kernels/norm.rs:226 -> avx/norm.rs:10-133):
 *   out = fused_quantized_linear[_residual](input, W.., apply_relu, res1, res2);  ln_out = layer_norm(out, ln_scale, ln_bias, -1, epsilon)
 * bit for bit the two calls (res1 / res2 may be NULL).  With K = 512 and N = 512 over a batch a workgroup of the GEMM holds whole
 * rows of the result and normalises them in its epilogue: one launch and one read of the sum less per transformer half-layer. */
int lele_hip_fused_quantized_linear_residual_ln(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8,
                                                const LeleTensor* weight_scale, const LeleTensor* weight_zero, const LeleTensor* bias,
                                                int apply_relu, const LeleTensor* res1, const LeleTensor* res2, const LeleTensor* ln_scale,
                                                const LeleTensor* ln_bias, float epsilon, LeleBuf* out, LeleBuf* ln_out, int64_t* out_shape,
                                                int32_t* out_rank);
/* the output half of a SAN-M attention block (SenseVoice's encoder layer; an FSMN memory block beside the attention) as one call:
 *   mem = depthwise_conv1d_tlc(v_src, x_offset, fsmn_w, fsmn_bias, pad_left, pad_right, relu = 0, add_input = 1)
 *   out = fused_quantized_linear_residual(input, W.., apply_relu, mem, res2);  ln_out = layer_norm(out, ln_scale, ln_bias, -1, epsilon)
 * (conv1d.rs:837 between two transposes + an Add, quantization.rs:77, two Adds, norm.rs:226) bit for bit those three calls; res2 and
 * fsmn_bias may be NULL.  With K = N = 512 over a batch the GEMM's workgroup computes the memory block of its 32 rows from a window of
 * v staged in LDS and normalises the rows in its epilogue -- three launches become one; other shapes issue the three calls. */
int lele_hip_sanm_out_block(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8, const LeleTensor* weight_scale,
                            const LeleTensor* weight_zero, const LeleTensor* bias, int apply_relu, const LeleTensor* v_src,
                            const LeleTensor* fsmn_w, const LeleTensor* fsmn_bias, int64_t x_offset, int64_t pad_left, int64_t pad_right,
                            const LeleTensor* res2, const LeleTensor* ln_scale, const LeleTensor* ln_bias, float epsilon, LeleBuf* out,
                            LeleBuf* ln_out, int64_t* out_shape, int32_t* out_rank);
/* two quantised linears with a ReLU between them (a transformer layer's feed-forward block):
 *   fused_quantized_linear[_residual](fused_quantized_linear(input, w1.., apply_relu = 1), w2.., apply_relu2, res1, res2)
 * (quantization.rs:77-169 twice; res1 / res2 may be NULL).  When the hidden layer is large its f32 tensor is never stored: the first
 * product runs twice on the matrix cores (once for the range, once quantising straight to the i8 rows the second product reads) */
int lele_hip_fused_ffn_quantized(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* w1_int8, const LeleTensor* w1_scale,
                                 const LeleTensor* w1_zero, const LeleTensor* b1, const LeleTensor* w2_int8,
                                 const LeleTensor* w2_scale, const LeleTensor* w2_zero, const LeleTensor* b2, int apply_relu2,
                                 const LeleTensor* res1, const LeleTensor* res2, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);
/* the same followed by the LayerNorm over the last axis that reads the result (the NEXT half-layer's LayerNorm in a transformer stack):
 *   out = fused_ffn_quantized(...);  ln_out = layer_norm(out, ln_scale, ln_bias, -1, epsilon)
 * bit for bit the two calls.  With 2048 hidden and 512 output columns over a batch the second product runs one 32-row tile a workgroup
 * with all columns (weights streamed from L2) and normalises its rows in the epilogue: the LayerNorm launch and one round trip of the sum go. */
int lele_hip_fused_ffn_quantized_ln(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* w1_int8, const LeleTensor* w1_scale,
                                    const LeleTensor* w1_zero, const LeleTensor* b1, const LeleTensor* w2_int8, const LeleTensor* w2_scale,
                                    const LeleTensor* w2_zero, const LeleTensor* b2, int apply_relu2, const LeleTensor* res1,
                                    const LeleTensor* res2, const LeleTensor* ln_scale, const LeleTensor* ln_bias, float epsilon, LeleBuf* out,
                                    LeleBuf* ln_out, int64_t* out_shape, int32_t* out_rank);
/* softmax(x * scale[0]) over the last axis: `mul` by a one-element tensor followed by `softmax` (norm.rs:8) */
int lele_hip_softmax_scaled(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, int32_t axis, LeleBuf* out,
                            int64_t* out_shape, int32_t* out_rank);
/* sqrt(pow(x[.., lo_start:lo_end, ..], exp_lo) + pow(x[.., hi_start:hi_end, ..], exp_hi)) along `axis` (bounds as Slice's: negative
 * from the end, clamped; the two ranges must be equally long): `slice`, `pow`, `slice`, `pow`, `add`, `sqrt` (manipulation.rs:209,
 * math.rs:1481, 414, 1460) in one pass -- the magnitude of a spectrum stored as [re | im] channel halves (Silero's
 * STFT-as-convolution).  exp_lo / exp_hi are one-element host constants. */
int lele_hip_halves_pow_add_sqrt(LeleCtx* ctx, const LeleTensor* x, int32_t axis, int64_t lo_start, int64_t lo_end, int64_t hi_start,
                                 int64_t hi_end, const LeleTensor* exp_lo, const LeleTensor* exp_hi, LeleBuf* out, int64_t* out_shape,
                                 int32_t* out_rank);
/* (a + b) + c on equal shapes: two consecutive `add`s (math.rs:414) */
int lele_hip_add3(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, const LeleTensor* c, LeleBuf* out, int64_t* out_shape,
                  int32_t* out_rank);
/* Transpose(0,2,1) -> depthwise conv1d (group = C, stride 1, dilation 1, conv1d.rs:837) -> Transpose(0,2,1), without the
 * transposes.  x f32 [B, T, P]: the C = w.shape[0] channels [x_offset, x_offset + C) of the last dimension are convolved
 * along T (x_offset = 0 and P = C for a plain [B, T, C] tensor; a non-zero offset reads one part of a packed projection in
 * place); w [C, 1, K] (K in 3, 5, 7, 11), bias [C] or NULL -> out [B, T + pad_left + pad_right - K + 1, C].
 * add_input: out += the convolved input itself (the FSMN "memory + input" Add that follows; needs equal lengths). */
int lele_hip_depthwise_conv1d_tlc(LeleCtx* ctx, const LeleTensor* x, int64_t x_offset, const LeleTensor* w, const LeleTensor* bias,
                                  int64_t pad_left, int64_t pad_right, int relu, int add_input, LeleBuf* out, int64_t* out_shape,
                                  int32_t* out_rank);
/* Batched matmul whose operands and result are STRIDED VIEWS (heads inside a packed [B, T, 3*D] projection; the result
 * stored straight into the [B, T, H, Dh] layout): the same MFMA kernels and tile choice as lele_hip_matmul, so the values
 * are bit-identical to copying the views out (slice / reshape / transpose) and calling matmul.  Element (bo, bi, r, c) of a
 * view lives at data[offset + bo*stride_outer + bi*stride_inner + r*stride_row + c*stride_col] (in elements); A is [m, k],
 * B is [k, n], the result [m, n]; A and B need unit stride along one of their two dimensions, the result along n.
 * out_dims is the shape reported for `out` (batch_outer*batch_inner*m*n elements). */
typedef struct LeleMatView {
    int64_t offset, stride_outer, stride_inner, stride_row, stride_col;
} LeleMatView;
int lele_hip_matmul_view(LeleCtx* ctx, const LeleTensor* a, const LeleMatView* a_view, const LeleTensor* b, const LeleMatView* b_view,
                         int64_t batch_outer, int64_t batch_inner, int64_t m, int64_t k, int64_t n, const LeleMatView* out_view,
                         const int64_t* out_dims, int32_t out_dims_rank, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* softmax(Q K^T * scale) V in one launch: lele_hip_matmul_view(Q view, K^T view) -> lele_hip_softmax_scaled ->
 * lele_hip_matmul_view(P, V view, out view), with the [.., T, T] score and probability tensors kept on chip.  Views as in
 * lele_hip_matmul_view: q is [t_q, dh], k is the B operand of the first product, i.e. K TRANSPOSED [dh, t_k], v is [t_k, dh],
 * the result [t_q, dh].  scale: one f32 value or NULL.  Same arithmetic as the sequence (f32 MFMA with the tiled GEMM's k order,
 * the row softmax of norm.rs:8), so batched results carry the sequence's bits.  Supported: dh == 128, t_k <= 512, unit stride
 * along dh for q / k / v / out, 16-byte aligned q / k rows -- anything else returns an error and the caller issues the sequence. */
int lele_hip_attention_view(LeleCtx* ctx, const LeleTensor* q, const LeleMatView* q_view, const LeleTensor* k, const LeleMatView* k_view,
                            const LeleTensor* v, const LeleMatView* v_view, int64_t batch_outer, int64_t batch_inner, int64_t t_q,
                            int64_t t_k, int64_t dh, const LeleTensor* scale_or_null, const LeleMatView* out_view,
                            const int64_t* out_dims, int32_t out_dims_rank, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

#ifdef __cplusplus
}
#endif
#endif /* LELE_HIP_H */
