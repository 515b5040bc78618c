// features_ops.hip -- the small feature operators: LFR, CMVN, generic radix-2 FFT (RealFft / ONNX STFT).
//
//   lele_hip_lfr                  <- /root/reference/src/features/lfr.rs:18-54      (pure gather, bit-exact)
//   lele_hip_cmvn(_apply_...)     <- /root/reference/src/features/cmvn.rs:14-92     (bit-exact: one thread per
//                                    feature dimension walks time in order, exactly like the two scalar passes)
//   lele_hip_rfft / stft / stft_power_spectrum
//                                 <- /root/reference/src/kernels/fft.rs:51-266, src/kernels/math.rs:2304-2439
//                                    same butterfly network and roundings as the x86 branch (see fe_core.h),
//                                    any power-of-two n_fft <= 4096, one workgroup per frame, stages through LDS.
#include "common.h"
#include "fe_core.h"

#include <math.h>

using namespace lele;

namespace {

// ---------------------------------------------------------------------------------------------- LFR
__global__ void lfr_kernel(const float* __restrict__ x, int64_t t, int64_t d, int64_t m, int64_t n, int64_t t_lfr,
                           float* __restrict__ out) {
    const int64_t d_out = d * m;
    const int64_t total = t_lfr * d_out;
    const int64_t pad = (m - 1) / 2;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / d_out, rem = idx - i * d_out;
        const int64_t block = rem / d, k = rem - block * d;
        int64_t raw = i * n + block - pad;
        raw = raw < 0 ? 0 : (raw > t - 1 ? t - 1 : raw);  // lfr.rs:42-43
        out[idx] = x[raw * d + k];
    }
}

// ---------------------------------------------------------------------------------------------- CMVN
// Statistics: one thread per feature dimension, coalesced across dimensions, SEQUENTIAL over time exactly as
// cmvn.rs:28-50 accumulates (sum and sum of squares in frame order).  The loads of 8 frames are issued together before
// the 8 dependent adds.  grid.y = utterance.  The normalisation itself is a separate fully parallel pass.
__global__ void cmvn_moments_kernel(const float* __restrict__ x, int64_t t, int64_t d, float eps,
                                    float* __restrict__ mean_out, float* __restrict__ sd_out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= d) return;
    const float* xu = x + (int64_t)blockIdx.y * t * d + k;
    float sum = 0.0f, sq = 0.0f;
    int64_t ti = 0;
    for (; ti + 8 <= t; ti += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xu[(ti + u) * d];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            sum = sum + v[u];
            sq = sq + v[u] * v[u];  // contraction is off: product rounded, then added
        }
    }
    for (; ti < t; ++ti) {
        const float v = xu[ti * d];
        sum = sum + v;
        sq = sq + v * v;
    }
    const float tf = (float)t;
    const float mean = sum / tf;
    float var = sq / tf - mean * mean;
    var = (var > 0.0f) ? var : 0.0f;  // f32::max(0.0) (NaN -> 0)
    mean_out[(int64_t)blockIdx.y * d + k] = mean;
    sd_out[(int64_t)blockIdx.y * d + k] = sqrtf(var + eps);
}
__global__ void cmvn_apply_kernel(const float* __restrict__ x, int64_t t, int64_t d, const float* __restrict__ mean,
                                  const float* __restrict__ sd, float* __restrict__ out) {
    const int64_t total = t * d, u = blockIdx.y;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = idx % d;
        out[u * total + idx] = (x[u * total + idx] - mean[u * d + k]) / sd[u * d + k];  // cmvn.rs:58-62
    }
}

__global__ void cmvn_stats_kernel(const float* __restrict__ x, int64_t t, int64_t d, float eps,
                                  const float* __restrict__ mean, const float* __restrict__ sd,
                                  float* __restrict__ out) {
    const int64_t total = t * d;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = idx % d;
        out[idx] = (x[idx] - mean[k]) / (sd[k] + eps);  // cmvn.rs:88
    }
}

// ---------------------------------------------------------------------------------------------- FFT
// One workgroup per frame.  LDS: re[n], im[n].  Butterfly network, twiddles and roundings are those of
// rfft_forward_f32_precomputed_avx2 (fft.rs:172-266): half_size >= 4 -> fma form, half_size < 4 -> scalar form.
// mode 0: complex output [frames, n/2+1, 2]; mode 1: power [frames, n/2+1]; mode 2: split re / im planes.
__global__ void fft_frames_kernel(const float* __restrict__ signal, int64_t sig_len, int64_t frame_stride,
                                  int64_t hop, int n, int log2n, int win_length, const float* __restrict__ window,
                                  const float* __restrict__ tw_re, const float* __restrict__ tw_im, int frames_per_row,
                                  int mode, float* __restrict__ out0, float* __restrict__ out1) {
    extern __shared__ float lds[];
    float* re = lds;
    float* im = lds + n;
    const int frame = blockIdx.x;
    const int row = frame / frames_per_row, fr = frame - row * frames_per_row;
    const float* sig = signal + (int64_t)row * frame_stride;
    const int64_t start = (int64_t)fr * hop;
    // frame_data[i] = signal[start+i] * window[i] (math.rs:2345-2351), scattered to the bit-reversed slot
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        float v = 0.0f;
        if (i < win_length && start + i < sig_len) {
            v = sig[start + i];
            if (window) v = v * window[i];
        }
        const int j = (int)(__brev((unsigned)i) >> (32 - log2n));
        re[j] = v;
        im[j] = 0.0f;
    }
    __syncthreads();
    int tw_off = 0;
    for (int size = 2; size <= n; size <<= 1) {
        const int half = size >> 1;
        for (int idx = threadIdx.x; idx < n / 2; idx += blockDim.x) {
            const int batch = idx / half, k = idx - batch * half;
            const int e = batch * size + k, o = e + half;
            const float wr = tw_re[tw_off + k], wi = tw_im[tw_off + k];
            float er = re[e], ei = im[e], orr = re[o], oi = im[o];
            if (half >= 4)
                fe::bfly_fma(wr, wi, er, ei, orr, oi);
            else
                fe::bfly_scalar(wr, wi, er, ei, orr, oi);
            re[e] = er;
            im[e] = ei;
            re[o] = orr;
            im[o] = oi;
        }
        tw_off += half;
        __syncthreads();
    }
    const int nb = n / 2 + 1;
    for (int k = threadIdx.x; k < nb; k += blockDim.x) {
        float r = re[k];
        float q = (k == 0 || k == nb - 1) ? 0.0f : im[k];  // fft.rs:256-261
        if (k == nb - 1) r = re[n / 2];
        const int64_t o = (int64_t)frame * nb + k;
        if (mode == 0) {
            out0[o * 2] = r;
            out0[o * 2 + 1] = q;
        } else if (mode == 1) {
            out0[o] = fe::power(r, q);
        } else {
            out0[o] = r;
            out1[o] = q;
        }
    }
}

static const float PI_F = 3.14159265358979323846264338327950288f;

struct FftTables {
    const float* tw_re;
    const float* tw_im;
};

// twiddle tables per n are immutable: cache them in the ctx weight map under a synthetic key
int get_fft_tables(LeleCtx* ctx, int64_t n, FftTables* out) {
    static const char tag_re = 0, tag_im = 0;
    auto key_re = std::make_tuple((const void*)&tag_re, (size_t)n, 101);
    auto key_im = std::make_tuple((const void*)&tag_im, (size_t)n, 102);
    auto it = ctx->weights.find(key_re);
    if (it == ctx->weights.end()) {
        std::vector<float> re, im;
        for (int64_t size = 2; size <= n; size *= 2) {  // fft.rs:136-157
            const int64_t half = size / 2, step = n / size;
            for (int64_t k = 0; k < half; ++k) {
                const float angle = -2.0f * PI_F * (float)(k * step) / (float)n;
                re.push_back(cosf(angle));
                im.push_back(sinf(angle));
            }
        }
        re.push_back(0.f);
        im.push_back(0.f);
        void *dre = nullptr, *dim = nullptr;
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMalloc(&dre, re.size() * 4));
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMalloc(&dim, im.size() * 4));
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMemcpy(dre, re.data(), re.size() * 4, hipMemcpyHostToDevice));
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMemcpy(dim, im.data(), im.size() * 4, hipMemcpyHostToDevice));
        ctx->weights[key_re] = dre;
        ctx->weights[key_im] = dim;
    }
    out->tw_re = (const float*)ctx->weights[key_re];
    out->tw_im = (const float*)ctx->weights[key_im];
    return 0;
}

int ilog2(int64_t n) {
    int l = 0;
    while ((int64_t(1) << (l + 1)) <= n) ++l;
    return l;
}

int launch_fft(LeleCtx* ctx, const float* sig, int64_t rows, int64_t sig_len, int64_t row_stride, int64_t hop,
               int64_t n_fft, int64_t win_length, const float* window, int64_t frames_per_row, int mode, float* out0,
               float* out1) {
    LELE_REQUIRE(n_fft >= 2 && (n_fft & (n_fft - 1)) == 0, "fft: length %lld is not a power of two", (long long)n_fft);
    LELE_REQUIRE(n_fft <= 4096, "fft: n_fft=%lld exceeds the device limit of 4096", (long long)n_fft);
    FftTables tb;
    LELE_TRY(get_fft_tables(ctx, n_fft, &tb));
    const int64_t frames = rows * frames_per_row;
    if (frames == 0) return 0;
    int threads = (int)std::min<int64_t>(256, std::max<int64_t>(64, n_fft / 2));
    hipLaunchKernelGGL(fft_frames_kernel, dim3((unsigned)frames), dim3(threads), (size_t)n_fft * 8, ctx->stream, sig,
                       sig_len, row_stride, hop, (int)n_fft, ilog2(n_fft), (int)win_length, window, tb.tw_re, tb.tw_im,
                       (int)frames_per_row, mode, out0, out1);
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

int stft_common(LeleCtx* ctx, const LeleTensor* signal, int64_t n_fft, int64_t hop, int64_t win_length,
                const LeleTensor* window, LeleBuf* out, int64_t* out_shape, int32_t* out_rank, bool power) {
    LELE_REQUIRE(ctx && signal && out, "stft: NULL argument");
    LELE_REQUIRE(signal->dtype == LELE_F32, "stft: signal must be f32");
    LELE_REQUIRE(win_length <= n_fft && hop > 0 && win_length > 0, "stft: need 0 < win_length <= n_fft and hop > 0");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t len = numel(signal);  // the reference treats input.data as ONE signal (math.rs:2312)
    const int64_t nfreq = n_fft / 2 + 1;
    if (len == 0) {  // math.rs:2313-2316
        out->bytes = 0;
        if (power) return set_shape(out_shape, out_rank, {0, 0, nfreq});
        return set_shape(out_shape, out_rank, {0, 0, nfreq, 2});
    }
    const int64_t frames = len < win_length ? 1 : (len - win_length) / hop + 1;
    LELE_TRY(ctx->arena_reset());
    const void* dsig = nullptr;
    LELE_TRY(ctx->dev_ptr(signal, &dsig));
    const float* dwin = nullptr;
    if (window) {
        LELE_REQUIRE(numel(window) >= win_length, "stft: window shorter than win_length");
        const void* w = nullptr;
        LELE_TRY(ctx->dev_ptr(window, &w));
        dwin = (const float*)w;
    } else {  // periodic Hann over win_length, math.rs:2328-2332
        std::vector<float> w(win_length);
        for (int64_t i = 0; i < win_length; ++i) w[i] = 0.5f * (1.0f - cosf(2.0f * PI_F * (float)i / (float)win_length));
        void* d = nullptr;
        LELE_TRY(ctx->arena_alloc(win_length * 4, &d));
        LELE_HIP_CHECK(hipMemcpyAsync(d, w.data(), win_length * 4, hipMemcpyHostToDevice, ctx->stream));
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipStreamSynchronize(ctx->stream));  // w is a local
        dwin = (const float*)d;
    }
    LELE_TRY(out->reserve((size_t)frames * nfreq * (power ? 1 : 2) * 4));
    LELE_TRY(launch_fft(ctx, (const float*)dsig, 1, len, 0, hop, n_fft, win_length, dwin, frames, power ? 1 : 0,
                        (float*)out->data, nullptr));
    // shape rule of math.rs:2362-2369
    int64_t batch = 1;
    for (int i = 0; i + 1 < signal->rank; ++i) batch *= signal->shape[i];
    if (signal->rank <= 1) {
        if (power) return set_shape(out_shape, out_rank, {frames, nfreq});
        return set_shape(out_shape, out_rank, {frames, nfreq, 2});
    }
    if (power) return set_shape(out_shape, out_rank, {batch, frames, nfreq});
    return set_shape(out_shape, out_rank, {batch, frames, nfreq, 2});
}

int feature_2d(const LeleTensor* x, int64_t* t, int64_t* d, const char* who) {
    if (x->rank == 2) {
        *t = x->shape[0];
        *d = x->shape[1];
        return 0;
    }
    if (x->rank == 3 && x->shape[0] == 1) {
        *t = x->shape[1];
        *d = x->shape[2];
        return 0;
    }
    set_error("%s expects [T, D] or [1, T, D] input", who);  // lfr.rs:25-28, cmvn.rs:22
    return 2;
}

}  // namespace

namespace lele {
// the per-stage twiddle tables of the radix-2 network (fft.rs:136-157), on the device: [n - 1] values each, stage by stage
int fft_twiddles(LeleCtx* ctx, int64_t n, const float** tw_re, const float** tw_im) {
    FftTables t;
    LELE_TRY(get_fft_tables(ctx, n, &t));
    *tw_re = t.tw_re;
    *tw_im = t.tw_im;
    return 0;
}
int fft_rows_power(LeleCtx* ctx, const float* rows_in, int64_t rows, int64_t n_fft, float* out_power) {
    return launch_fft(ctx, rows_in, rows, n_fft, n_fft, n_fft, n_fft, n_fft, nullptr, 1, 1, out_power, nullptr);
}
}  // namespace lele

extern "C" {

int lele_hip_lfr(LeleCtx* ctx, const LeleTensor* x, int64_t m, int64_t n, LeleBuf* out, int64_t* out_shape,
                 int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out && m > 0 && n > 0, "lfr: bad argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t t, d;
    LELE_TRY(feature_2d(x, &t, &d, "LFR"));
    if (t == 0) {
        out->bytes = 0;
        return set_shape(out_shape, out_rank, {0, d * m});
    }
    const int64_t t_lfr = (t + n - 1) / n;
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)t_lfr * d * m * 4));
    const int64_t total = t_lfr * d * m;
    const int blocks = (int)std::min<int64_t>((total + 255) / 256, 2048);
    hipLaunchKernelGGL(lfr_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)dx, t, d, m, n, t_lfr,
                       (float*)out->data);
    LELE_HIP_CHECK(hipGetLastError());
    return set_shape(out_shape, out_rank, {t_lfr, d * m});
}

int lele_hip_cmvn(LeleCtx* ctx, const LeleTensor* x, float eps, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "cmvn: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t t, d, nb = 1;
    if (x->rank == 3 && x->shape[0] > 1) {  // extension: [B, T, D] = B utterances, each normalised with its own statistics
        nb = x->shape[0];
        t = x->shape[1];
        d = x->shape[2];
    } else {
        LELE_TRY(feature_2d(x, &t, &d, "CMVN"));
    }
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)nb * t * d * 4));
    if (t > 0 && d > 0) {
        void *dm = nullptr, *ds = nullptr;
        LELE_TRY(ctx->arena_alloc((size_t)nb * d * 4, &dm));
        LELE_TRY(ctx->arena_alloc((size_t)nb * d * 4, &ds));
        hipLaunchKernelGGL(cmvn_moments_kernel, dim3((unsigned)((d + 63) / 64), (unsigned)nb), dim3(64), 0, ctx->stream,
                           (const float*)dx, t, d, eps, (float*)dm, (float*)ds);
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((t * d + 255) / 256, 1024));
        hipLaunchKernelGGL(cmvn_apply_kernel, dim3(blocks, (unsigned)nb), dim3(256), 0, ctx->stream, (const float*)dx, t, d,
                           (const float*)dm, (const float*)ds, (float*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    std::vector<int64_t> shp(x->shape, x->shape + x->rank);
    return set_shape_v(out_shape, out_rank, shp);
}

int lele_hip_cmvn_apply_with_stats(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* mean, const LeleTensor* sd,
                                   float eps, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && mean && sd && out, "cmvn_apply_with_stats: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t t, d;
    LELE_TRY(feature_2d(x, &t, &d, "CMVN"));
    LELE_REQUIRE(numel(mean) == d, "Mean dimension mismatch");  // cmvn.rs:81
    LELE_REQUIRE(numel(sd) == d, "Std dimension mismatch");     // cmvn.rs:82
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dm = nullptr, *ds = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(mean, &dm));
    LELE_TRY(ctx->dev_ptr(sd, &ds));
    LELE_TRY(out->reserve((size_t)t * d * 4));
    if (t * d > 0) {
        const int blocks = (int)std::min<int64_t>((t * d + 255) / 256, 2048);
        hipLaunchKernelGGL(cmvn_stats_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)dx, t, d, eps,
                           (const float*)dm, (const float*)ds, (float*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    std::vector<int64_t> shp(x->shape, x->shape + x->rank);
    return set_shape_v(out_shape, out_rank, shp);
}

int lele_hip_rfft(LeleCtx* ctx, const LeleTensor* x, LeleBuf* out_re, LeleBuf* out_im, int64_t* out_shape,
                  int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out_re && out_im, "rfft: NULL argument");
    LELE_REQUIRE(x->dtype == LELE_F32 && x->rank >= 1, "rfft: x must be f32 [rows, n]");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = x->shape[x->rank - 1];
    const int64_t rows = n ? numel(x) / n : 0;
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    const int64_t nb = n / 2 + 1;
    LELE_TRY(out_re->reserve((size_t)rows * nb * 4));
    LELE_TRY(out_im->reserve((size_t)rows * nb * 4));
    LELE_TRY(launch_fft(ctx, (const float*)dx, rows, n, n, n, n, n, nullptr, 1, 2, (float*)out_re->data,
                        (float*)out_im->data));
    return set_shape(out_shape, out_rank, {rows, nb});
}

int lele_hip_stft(LeleCtx* ctx, const LeleTensor* signal, int64_t n_fft, int64_t hop, int64_t win_length,
                  const LeleTensor* window, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    return stft_common(ctx, signal, n_fft, hop, win_length, window, out, out_shape, out_rank, false);
}
int lele_hip_stft_power_spectrum(LeleCtx* ctx, const LeleTensor* signal, int64_t n_fft, int64_t hop,
                                 int64_t win_length, const LeleTensor* window, LeleBuf* out, int64_t* out_shape,
                                 int32_t* out_rank) {
    return stft_common(ctx, signal, n_fft, hop, win_length, window, out, out_shape, out_rank, true);
}

}  // extern "C"
