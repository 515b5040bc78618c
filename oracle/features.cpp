// oracle/features.cpp -- CPU restatement of lele's audio front-end (TEST INFRASTRUCTURE, see oracle.h).
//
// Follows, statement by statement, /root/reference:
//   src/kernels/fft.rs:2-49      rfft_forward_f32 (twiddles on the fly)
//   src/kernels/fft.rs:79-134    rfft_forward_f32_precomputed_scalar
//   src/kernels/fft.rs:136-170   precompute_twiddles, bit_reverse
//   src/kernels/fft.rs:172-266   rfft_forward_f32_precomputed_avx2 (same intrinsics via <immintrin.h>)
//   src/features/window.rs:2-13  hann_window (symmetric)
//   src/features/mel.rs:1-128    hz_to_mel_htk, mel_to_hz_htk, mel_filterbank, SparseMelBank, log_compress
//   src/features/pipeline.rs:38-193  SenseVoiceFrontend::{new,compute}
//   src/features/lfr.rs:18-54    Lfr::compute
//   src/features/cmvn.rs:14-92   Cmvn::{compute,apply_with_stats}
//   src/kernels/math.rs:2304-2439 stft, stft_power_spectrum
//
// Build with -ffp-contract=off: Rust never contracts a*b+c, FMA appears only where the reference
// writes _mm256_fmadd/fmsub explicitly.
#include "oracle.h"

#include <immintrin.h>
#include <math.h>
#include <string.h>

#include <vector>

static const float PI_F = 3.14159265358979323846264338327950288f;  // core::f32::consts::PI

static inline int64_t bit_reverse(int64_t n, int log2n) {  // fft.rs:160-169
    int64_t r = 0, x = n;
    for (int i = 0; i < log2n; ++i) {
        r = (r << 1) | (x & 1);
        x >>= 1;
    }
    return r;
}
static inline int ilog2(int64_t n) {
    int l = 0;
    while ((int64_t(1) << (l + 1)) <= n) ++l;
    return l;
}

extern "C" void orc_hann_window(int64_t size, float* out) {  // window.rs:2-13
    if (size == 0) return;
    if (size == 1) {
        out[0] = 1.0f;
        return;
    }
    for (int64_t n = 0; n < size; ++n) {
        // 0.5 * (1.0 - (2.0 * PI * n as f32 / (size - 1) as f32).cos())
        float arg = 2.0f * PI_F * (float)n / (float)(size - 1);
        out[n] = 0.5f * (1.0f - cosf(arg));
    }
}

extern "C" void orc_precompute_twiddles(int64_t n, float* tw_re, float* tw_im, int64_t* bit_rev) {
    int log2n = ilog2(n);
    for (int64_t i = 0; i < n; ++i) bit_rev[i] = bit_reverse(i, log2n);
    int64_t off = 0;
    for (int64_t size = 2; size <= n; size *= 2) {
        int64_t half_size = size / 2, step = n / size;
        for (int64_t k = 0; k < half_size; ++k) {
            // -2.0 * PI * (k * step) as f32 / n as f32
            float angle = -2.0f * PI_F * (float)(k * step) / (float)n;
            tw_re[off + k] = cosf(angle);
            tw_im[off + k] = sinf(angle);
        }
        off += half_size;
    }
}

static void rfft_onthefly(const float* input, int64_t n, float* out_re, float* out_im) {  // fft.rs:2-49
    int log2n = ilog2(n);
    int64_t half = n / 2 + 1;
    std::vector<float> re(n, 0.0f), im(n, 0.0f);
    for (int64_t i = 0; i < n; ++i) re[bit_reverse(i, log2n)] = input[i];
    for (int64_t size = 2; size <= n; size *= 2) {
        int64_t half_size = size / 2, step = n / size, num_batches = n / size;
        for (int64_t batch = 0; batch < num_batches; ++batch) {
            int64_t bs = batch * size;
            for (int64_t k = 0; k < half_size; ++k) {
                int64_t e = bs + k, o = bs + half_size + k;
                float angle = -2.0f * PI_F * (float)(k * step) / (float)n;
                float wr = cosf(angle), wi = sinf(angle);
                float tr = wr * re[o] - wi * im[o];
                float ti = wr * im[o] + wi * re[o];
                re[o] = re[e] - tr;
                im[o] = im[e] - ti;
                re[e] += tr;
                im[e] += ti;
            }
        }
    }
    out_re[0] = re[0];
    out_im[0] = 0.0f;
    if (half > 1) {
        out_re[half - 1] = re[n / 2];
        out_im[half - 1] = 0.0f;
    }
    for (int64_t k = 1; k < half - 1; ++k) {
        out_re[k] = re[k];
        out_im[k] = im[k];
    }
}

static void rfft_pre_scalar(const float* input, int64_t n, const float* twr, const float* twi,
                            const int64_t* br, float* re, float* im, float* out_re, float* out_im) {
    int64_t half = n / 2 + 1;  // fft.rs:79-134
    for (int64_t i = 0; i < n; ++i) re[br[i]] = input[i];
    for (int64_t i = 0; i < n; ++i) im[i] = 0.0f;
    int64_t tw_off = 0;
    for (int64_t size = 2; size <= n; size *= 2) {
        int64_t half_size = size / 2, num_batches = n / size;
        for (int64_t batch = 0; batch < num_batches; ++batch) {
            int64_t base = batch * size;
            for (int64_t k = 0; k < half_size; ++k) {
                int64_t e = base + k, o = base + half_size + k;
                float wr = twr[tw_off + k], wi = twi[tw_off + k];
                float odd_re = re[o], odd_im = im[o];
                float tr = wr * odd_re - wi * odd_im;
                float ti = wr * odd_im + wi * odd_re;
                re[o] = re[e] - tr;
                im[o] = im[e] - ti;
                re[e] += tr;
                im[e] += ti;
            }
        }
        tw_off += half_size;
    }
    out_re[0] = re[0];
    out_im[0] = 0.0f;
    if (half > 1) {
        out_re[half - 1] = re[n / 2];
        out_im[half - 1] = 0.0f;
    }
    for (int64_t k = 1; k < half - 1; ++k) {
        out_re[k] = re[k];
        out_im[k] = im[k];
    }
}

// fft.rs:172-266 -- identical intrinsic sequence (fmsub/fmadd for the twiddle product, plain add/sub after).
static void rfft_pre_avx2(const float* input, int64_t n, const float* twr, const float* twi,
                          const int64_t* br, float* re, float* im, float* out_re, float* out_im) {
    int64_t half = n / 2 + 1;
    for (int64_t i = 0; i < n; ++i) re[br[i]] = input[i];
    for (int64_t i = 0; i < n; ++i) im[i] = 0.0f;
    int64_t tw_off = 0;
    for (int64_t size = 2; size <= n; size *= 2) {
        int64_t half_size = size / 2, num_batches = n / size;
        for (int64_t batch = 0; batch < num_batches; ++batch) {
            int64_t base = batch * size;
            int64_t k = 0;
            while (k + 8 <= half_size) {
                __m256 ev_re = _mm256_loadu_ps(re + base + k);
                __m256 ev_im = _mm256_loadu_ps(im + base + k);
                __m256 od_re = _mm256_loadu_ps(re + base + half_size + k);
                __m256 od_im = _mm256_loadu_ps(im + base + half_size + k);
                __m256 wr = _mm256_loadu_ps(twr + tw_off + k);
                __m256 wi = _mm256_loadu_ps(twi + tw_off + k);
                __m256 t_re = _mm256_fmsub_ps(wr, od_re, _mm256_mul_ps(wi, od_im));
                __m256 t_im = _mm256_fmadd_ps(wr, od_im, _mm256_mul_ps(wi, od_re));
                _mm256_storeu_ps(re + base + half_size + k, _mm256_sub_ps(ev_re, t_re));
                _mm256_storeu_ps(im + base + half_size + k, _mm256_sub_ps(ev_im, t_im));
                _mm256_storeu_ps(re + base + k, _mm256_add_ps(ev_re, t_re));
                _mm256_storeu_ps(im + base + k, _mm256_add_ps(ev_im, t_im));
                k += 8;
            }
            while (k + 4 <= half_size) {
                __m128 ev_re = _mm_loadu_ps(re + base + k);
                __m128 ev_im = _mm_loadu_ps(im + base + k);
                __m128 od_re = _mm_loadu_ps(re + base + half_size + k);
                __m128 od_im = _mm_loadu_ps(im + base + half_size + k);
                __m128 wr = _mm_loadu_ps(twr + tw_off + k);
                __m128 wi = _mm_loadu_ps(twi + tw_off + k);
                __m128 t_re = _mm_fmsub_ps(wr, od_re, _mm_mul_ps(wi, od_im));
                __m128 t_im = _mm_fmadd_ps(wr, od_im, _mm_mul_ps(wi, od_re));
                _mm_storeu_ps(re + base + half_size + k, _mm_sub_ps(ev_re, t_re));
                _mm_storeu_ps(im + base + half_size + k, _mm_sub_ps(ev_im, t_im));
                _mm_storeu_ps(re + base + k, _mm_add_ps(ev_re, t_re));
                _mm_storeu_ps(im + base + k, _mm_add_ps(ev_im, t_im));
                k += 4;
            }
            while (k < half_size) {
                int64_t e = base + k, o = base + half_size + k;
                float wr = twr[tw_off + k], wi = twi[tw_off + k];
                float odd_re = re[o], odd_im = im[o];
                float tr = wr * odd_re - wi * odd_im;
                float ti = wr * odd_im + wi * odd_re;
                re[o] = re[e] - tr;
                im[o] = im[e] - ti;
                re[e] += tr;
                im[e] += ti;
                k += 1;
            }
        }
        tw_off += half_size;
    }
    out_re[0] = re[0];
    out_im[0] = 0.0f;
    if (half > 1) {
        out_re[half - 1] = re[n / 2];
        out_im[half - 1] = 0.0f;
    }
    for (int64_t k = 1; k < half - 1; ++k) {
        out_re[k] = re[k];
        out_im[k] = im[k];
    }
}

namespace {
struct RealFft {  // features/fft.rs:1-49 (RealFft::new + the call into kernels::fft)
    int64_t n;
    std::vector<float> tw_re, tw_im, re_buf, im_buf;
    std::vector<int64_t> bit_rev;
    explicit RealFft(int64_t len) : n(len), tw_re(len), tw_im(len), re_buf(len), im_buf(len), bit_rev(len) {
        orc_precompute_twiddles(len, tw_re.data(), tw_im.data(), bit_rev.data());
    }
    // x86_64 dispatch of rfft_forward_f32_precomputed (fft.rs:51-77): AVX2+FMA present on every host we run on.
    void forward(const float* in, float* out_re, float* out_im) {
        rfft_pre_avx2(in, n, tw_re.data(), tw_im.data(), bit_rev.data(), re_buf.data(), im_buf.data(), out_re, out_im);
    }
};
}  // namespace

extern "C" void orc_rfft(const float* input, int64_t n, float* out_re, float* out_im, int mode) {
    if (mode == 0) {
        rfft_onthefly(input, n, out_re, out_im);
        return;
    }
    std::vector<float> twr(n), twi(n), re(n), im(n);
    std::vector<int64_t> br(n);
    orc_precompute_twiddles(n, twr.data(), twi.data(), br.data());
    if (mode == 1)
        rfft_pre_scalar(input, n, twr.data(), twi.data(), br.data(), re.data(), im.data(), out_re, out_im);
    else
        rfft_pre_avx2(input, n, twr.data(), twi.data(), br.data(), re.data(), im.data(), out_re, out_im);
}

extern "C" float orc_hz_to_mel_htk(float hz) { return 2595.0f * log10f(1.0f + hz / 700.0f); }
extern "C" float orc_mel_to_hz_htk(float mel) { return 700.0f * (powf(10.0f, mel / 2595.0f) - 1.0f); }

extern "C" void orc_mel_filterbank(float sample_rate, int64_t n_fft, int64_t n_mels, float f_min, float f_max,
                                   float* weights) {  // mel.rs:7-45
    if (f_max < 0.0f) f_max = sample_rate / 2.0f;
    int64_t n_freqs = n_fft / 2 + 1;
    float mel_min = orc_hz_to_mel_htk(f_min), mel_max = orc_hz_to_mel_htk(f_max);
    int64_t mel_points = n_mels + 2;
    float mel_step = (mel_max - mel_min) / (float)(n_mels + 1);
    std::vector<float> hz(mel_points), fft_freqs(n_freqs);
    for (int64_t i = 0; i < mel_points; ++i) hz[i] = orc_mel_to_hz_htk(mel_min + (float)i * mel_step);
    for (int64_t i = 0; i < n_freqs; ++i) fft_freqs[i] = (float)i * sample_rate / (float)n_fft;
    for (int64_t i = 0; i < n_mels; ++i) {
        float f_left = hz[i], f_center = hz[i + 1], f_right = hz[i + 2];
        for (int64_t j = 0; j < n_freqs; ++j) {
            float f = fft_freqs[j], val = 0.0f;
            if (f > f_left && f < f_center)
                val = (f - f_left) / (f_center - f_left);
            else if (f >= f_center && f < f_right)
                val = (f_right - f) / (f_right - f_center);
            weights[i * n_freqs + j] = val;
        }
    }
}

namespace {
struct SparseMelBank {  // mel.rs:48-105
    int64_t n_mels, n_freqs;
    std::vector<int64_t> start;
    std::vector<std::vector<float>> w;
    SparseMelBank(float sr, int64_t n_fft, int64_t nm, float f_min, float f_max) : n_mels(nm), n_freqs(n_fft / 2 + 1) {
        std::vector<float> dense(n_mels * n_freqs);
        orc_mel_filterbank(sr, n_fft, n_mels, f_min, f_max, dense.data());
        for (int64_t i = 0; i < n_mels; ++i) {
            const float* row = dense.data() + i * n_freqs;
            int64_t s = 0;
            while (s < n_freqs && row[s] == 0.0f) ++s;
            int64_t e = n_freqs;
            while (e > s && row[e - 1] == 0.0f) --e;
            if (s < e) {
                start.push_back(s);
                w.emplace_back(row + s, row + e);
            } else {
                start.push_back(0);
                w.emplace_back();
            }
        }
    }
    void apply(const float* power, float* out) const {
        for (int64_t i = 0; i < n_mels; ++i) {
            float sum = 0.0f;
            const float* p = power + start[i];
            for (size_t j = 0; j < w[i].size(); ++j) sum += w[i][j] * p[j];
            out[i] = sum;
        }
    }
};
}  // namespace

extern "C" void orc_sparse_mel_apply(float sr, int64_t n_fft, int64_t n_mels, float f_min, float f_max,
                                     const float* power, float* out) {
    SparseMelBank(sr, n_fft, n_mels, f_min, f_max).apply(power, out);
}

extern "C" void orc_lfr(const float* in, int64_t t, int64_t d, int64_t m, int64_t n, float* out) {  // lfr.rs:18-54
    if (t == 0) return;
    int64_t t_lfr = (t + n - 1) / n, d_out = d * m, pad = (m - 1) / 2;
    for (int64_t i = 0; i < t_lfr; ++i) {
        int64_t start_frame = i * n;
        for (int64_t block = 0; block < m; ++block) {
            int64_t raw = start_frame + block - pad;
            int64_t c = raw < 0 ? 0 : (raw > t - 1 ? t - 1 : raw);
            memcpy(out + i * d_out + block * d, in + c * d, sizeof(float) * d);
        }
    }
}

extern "C" int64_t orc_frontend_shape(int64_t pcm_len, int64_t sample_rate, float frame_length_ms,
                                      float frame_shift_ms, int64_t lfr_n, int64_t* num_frames) {
    int64_t frame_len = (int64_t)((float)sample_rate * frame_length_ms / 1000.0f);  // pipeline.rs:39
    int64_t hop_len = (int64_t)((float)sample_rate * frame_shift_ms / 1000.0f);     // pipeline.rs:42
    if (pcm_len < frame_len) {
        if (num_frames) *num_frames = 0;
        return 0;
    }
    int64_t nf = (pcm_len - frame_len) / hop_len + 1;  // pipeline.rs:73
    if (num_frames) *num_frames = nf;
    return (nf + lfr_n - 1) / lfr_n;
}

extern "C" int64_t orc_frontend_compute(const float* pcm, int64_t pcm_len, int64_t sample_rate, int64_t n_mels,
                                        float frame_length_ms, float frame_shift_ms, int64_t lfr_m, int64_t lfr_n,
                                        float* mel_out, float* out) {
    int64_t frame_len = (int64_t)((float)sample_rate * frame_length_ms / 1000.0f);
    int64_t n_fft = frame_len > 400 ? 1024 : 512;  // pipeline.rs:40
    int64_t hop_len = (int64_t)((float)sample_rate * frame_shift_ms / 1000.0f);
    if (pcm_len < frame_len) return 0;  // TensorView::empty()
    int64_t num_frames = (pcm_len - frame_len) / hop_len + 1;

    std::vector<float> window(frame_len);
    orc_hann_window(frame_len, window.data());
    RealFft fft(n_fft);
    SparseMelBank bank((float)sample_rate, n_fft, n_mels, 20.0f, -1.0f);

    std::vector<float> mel_local;
    if (!mel_out) {
        mel_local.resize(num_frames * n_mels);
        mel_out = mel_local.data();
    }
    std::vector<float> frame_buf(n_fft, 0.0f), raw(frame_len), fre(n_fft / 2 + 1), fim(n_fft / 2 + 1),
        power(n_fft / 2 + 1);
    const float scale = 32768.0f, preemph = 0.97f;
    for (int64_t i = 0; i < num_frames; ++i) {
        const float* p = pcm + i * hop_len;
        for (int64_t j = 0; j < frame_len; ++j) raw[j] = p[j] * scale;     // 1. scale
        float sum = 0.0f;                                                     // 2. mean (sequential f32 sum)
        for (int64_t j = 0; j < frame_len; ++j) sum += raw[j];
        float mean = sum / (float)frame_len;
        for (int64_t j = 0; j < frame_len; ++j) raw[j] -= mean;
        for (int64_t j = frame_len - 1; j >= 1; --j) raw[j] -= preemph * raw[j - 1];  // 3. pre-emphasis, j>=1 only
        for (int64_t j = 0; j < frame_len; ++j) frame_buf[j] = raw[j] * window[j];    // 4. window
        for (int64_t j = frame_len; j < n_fft; ++j) frame_buf[j] = 0.0f;
        fft.forward(frame_buf.data(), fre.data(), fim.data());                        // 5. FFT (bins 0..n/2)
        for (int64_t j = 0; j < n_fft / 2 + 1; ++j) power[j] = fre[j] * fre[j] + fim[j] * fim[j];  // 6.
        float* mel = mel_out + i * n_mels;
        bank.apply(power.data(), mel);                                                // 7.
        for (int64_t j = 0; j < n_mels; ++j) mel[j] = logf(mel[j] > 1e-5f ? mel[j] : 1e-5f);  // 8. ln(max(x,eps))
    }
    orc_lfr(mel_out, num_frames, n_mels, lfr_m, lfr_n, out);
    return (num_frames + lfr_n - 1) / lfr_n;
}

extern "C" void orc_cmvn(const float* in, int64_t t, int64_t d, float eps, float* out) {  // cmvn.rs:14-66
    if (t == 0) return;
    std::vector<float> sums(d, 0.0f), sq(d, 0.0f), means(d), stds(d);
    for (int64_t ti = 0; ti < t; ++ti)
        for (int64_t k = 0; k < d; ++k) {
            float v = in[ti * d + k];
            sums[k] += v;
            sq[k] += v * v;
        }
    float tf = (float)t;
    for (int64_t k = 0; k < d; ++k) {
        float mean = sums[k] / tf;
        float var = sq[k] / tf - mean * mean;
        if (!(var > 0.0f)) var = 0.0f;  // f32::max(0.0): NaN -> 0.0
        means[k] = mean;
        stds[k] = sqrtf(var + eps);
    }
    for (int64_t ti = 0; ti < t; ++ti)
        for (int64_t k = 0; k < d; ++k) out[ti * d + k] = (in[ti * d + k] - means[k]) / stds[k];
}

extern "C" void orc_cmvn_apply_with_stats(const float* in, int64_t t, int64_t d, float eps, const float* mean,
                                          const float* std_, float* out) {  // cmvn.rs:67-92
    for (int64_t ti = 0; ti < t; ++ti)
        for (int64_t k = 0; k < d; ++k) out[ti * d + k] = (in[ti * d + k] - mean[k]) / (std_[k] + eps);
}

static int64_t stft_common(const float* signal, int64_t len, int64_t n_fft, int64_t hop, int64_t win_length,
                           const float* window, float* out, bool power) {
    if (len == 0) return 0;
    int64_t num_frames = len < win_length ? 1 : (len - win_length) / hop + 1;
    int64_t n_freqs = n_fft / 2 + 1;
    std::vector<float> w(win_length);
    if (window)
        memcpy(w.data(), window, sizeof(float) * win_length);
    else  // periodic Hann, math.rs:2328-2332
        for (int64_t i = 0; i < win_length; ++i)
            w[i] = 0.5f * (1.0f - cosf(2.0f * PI_F * (float)i / (float)win_length));
    RealFft fft(n_fft);
    std::vector<float> frame(n_fft), fre(n_freqs), fim(n_freqs);
    for (int64_t f = 0; f < num_frames; ++f) {
        int64_t start = f * hop;
        for (int64_t i = 0; i < n_fft; ++i)
            frame[i] = (i < win_length && start + i < len) ? signal[start + i] * w[i] : 0.0f;
        fft.forward(frame.data(), fre.data(), fim.data());
        for (int64_t k = 0; k < n_freqs; ++k) {
            if (power)
                out[f * n_freqs + k] = fre[k] * fre[k] + fim[k] * fim[k];
            else {
                out[(f * n_freqs + k) * 2] = fre[k];
                out[(f * n_freqs + k) * 2 + 1] = fim[k];
            }
        }
    }
    return num_frames;
}
extern "C" int64_t orc_stft(const float* s, int64_t len, int64_t n_fft, int64_t hop, int64_t wl, const float* w,
                            float* out) {
    return stft_common(s, len, n_fft, hop, wl, w, out, false);
}
extern "C" int64_t orc_stft_power(const float* s, int64_t len, int64_t n_fft, int64_t hop, int64_t wl,
                                  const float* w, float* out) {
    return stft_common(s, len, n_fft, hop, wl, w, out, true);
}
