// app.hip -- the application-side pre/post-processing steps that bracket a lele model run,
// moved to the device so that (a) 16-bit PCM crosses PCIe instead of f32 and (b) only token ids leave the GPU.
//
//   lele_hip_wav_to_f32    <- /root/reference/examples/sensevoice/src/audio.rs:52-73   (s16 / u8 -> f32, stereo -> mono)
//   lele_hip_argmax_last   <- /root/reference/examples/sensevoice/src/tokenizer.rs:50-61 (per-frame arg-max of the logits;
//                             Iterator::max_by keeps the LAST of equal maxima)
//   lele_hip_token_filter  <- tokenizer.rs:63-71  (drop blank / special ids, keep frame order; no CTC collapsing upstream)
//   lele_hip_image_preprocess   <- /root/reference/examples/yolo26n-seg/src/image.rs:62-111 (PIL-style nearest resize to
//                             target x target, HWC u8 -> CHW f32 / 255)
//   lele_hip_yolo_seg_postprocess <- image.rs:127-265 (score / box filter in query order, box rescale, per-detection mask =
//                             sigmoid(coeffs . mask_features), nearest upscale, box crop, thresholds)
// All exact: one IEEE operation per sample / pure comparisons / the reference's operation order; the only transcendental
// (the mask sigmoid's expf, libm upstream) is evaluated in double and rounded once, i.e. correctly rounded like glibc's.
#include "common.h"

using namespace lele;

namespace {

__global__ void wav_to_f32_kernel(const uint8_t* __restrict__ bytes, int64_t frames, int bits, int channels,
                                  float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < frames; i += (int64_t)gridDim.x * blockDim.x) {
        float s[2] = {0.0f, 0.0f};
        for (int c = 0; c < channels; ++c) {
            const int64_t k = i * channels + c;
            if (bits == 16) {
                const int16_t v = (int16_t)((uint16_t)bytes[2 * k] | ((uint16_t)bytes[2 * k + 1] << 8));  // i16::from_le_bytes
                s[c] = (float)v / 32768.0f;
            } else {
                s[c] = ((float)bytes[k] - 128.0f) / 128.0f;
            }
        }
        out[i] = channels == 2 ? (s[0] + s[1]) / 2.0f : s[0];
    }
}

// one workgroup per row; (value, index) pairs reduced with "greater value wins, equal values: greater index wins"
__global__ __launch_bounds__(256) void argmax_last_kernel(const float* __restrict__ x, int64_t rows, int64_t v,
                                                          int32_t* __restrict__ out) {
    const int64_t row = blockIdx.x;
    const float* p = x + row * v;
    float best = -INFINITY;
    int64_t bi = -1;
    for (int64_t j = threadIdx.x; j < v; j += 256) {
        const float val = p[j];
        if (bi < 0 || val >= best) {  // later index replaces an equal value
            best = val;
            bi = j;
        }
    }
    __shared__ float sv[256];
    __shared__ int64_t si[256];
    sv[threadIdx.x] = best;
    si[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            const float ov = sv[threadIdx.x + off];
            const int64_t oi = si[threadIdx.x + off];
            const float mv = sv[threadIdx.x];
            const int64_t mi = si[threadIdx.x];
            if (oi >= 0 && (mi < 0 || ov > mv || (ov == mv && oi > mi))) {
                sv[threadIdx.x] = ov;
                si[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[row] = (int32_t)(si[0] < 0 ? 0 : si[0]);  // unwrap_or(0) for empty rows
}

// one workgroup per batch row: ordered compaction of the ids whose skip flag is clear (ballot prefix inside a wave,
// LDS prefix across waves, a running base across 256-frame chunks)
__global__ __launch_bounds__(256) void token_filter_kernel(const int32_t* __restrict__ ids, int64_t t, const uint8_t* __restrict__ skip,
                                                           int64_t vocab, int32_t* __restrict__ out, int32_t* __restrict__ counts) {
    const int64_t row = blockIdx.x;
    const int32_t* p = ids + row * t;
    int32_t* o = out + row * t;
    __shared__ int wave_cnt[4];
    __shared__ int base;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int64_t c0 = 0; c0 < t; c0 += 256) {
        const int64_t j = c0 + threadIdx.x;
        const int32_t id = j < t ? p[j] : 0;
        // tokenizer.rs:63-69: ids beyond the vocabulary are dropped, id 0 (blank) and <|...|> specials are skipped
        const bool keep = j < t && id >= 0 && id < vocab && skip[id] == 0;
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_cnt[wave] = __popcll(m);
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wave; ++w) woff += wave_cnt[w];
        const int b0 = base;
        if (keep) o[b0 + woff + before] = id;
        __syncthreads();
        if (threadIdx.x == 0) base = b0 + wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int n = base;
    for (int64_t j = n + threadIdx.x; j < t; j += 256) o[j] = -1;
    if (threadIdx.x == 0) counts[row] = n;
}

// image.rs:84-105 + 69-79: dst(y, x) <- src(min(floor((y + 0.5) * H / T), H - 1), min(floor((x + 0.5) * W / T), W - 1)) / 255
__global__ void image_preprocess_kernel(const uint8_t* __restrict__ rgb, int h, int w, int target, float* __restrict__ out) {
    const int total = target * target;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / target, x = i - y * target;
        int sx = (int)floorf(((float)x + 0.5f) * (float)w / (float)target);
        int sy = (int)floorf(((float)y + 0.5f) * (float)h / (float)target);
        sx = sx < w - 1 ? sx : w - 1;
        sy = sy < h - 1 ? sy : h - 1;
        const uint8_t* px = rgb + ((int64_t)sy * w + sx) * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) out[(int64_t)c * total + i] = (float)px[c] / 255.0f;
    }
}

constexpr int kQueries = 300, kLogitLen = 38, kMaskDim = 32;  // image.rs:135-136,151

// image.rs:159-203: one workgroup an image (blockIdx.x); kept queries are written in query order, the rows behind them are zeros
__global__ __launch_bounds__(320) void yolo_dets_kernel(const float* __restrict__ logits_all, float img_w, float img_h, float threshold,
                                                        int num_classes, float* __restrict__ dets_all, int32_t* __restrict__ count_all) {
    __shared__ int wave_cnt[5];
    const float* logits = logits_all + (int64_t)blockIdx.x * kQueries * kLogitLen;
    float* dets = dets_all + (int64_t)blockIdx.x * kQueries * kLogitLen;
    int32_t* count = count_all + blockIdx.x;
    const int i = threadIdx.x, lane = i & 63, wave = i >> 6;
    const float* q = logits + (int64_t)(i < kQueries ? i : 0) * kLogitLen;
    const float score = q[4];
    const float x1r = q[0], y1r = q[1], x2r = q[2], y2r = q[3];
    const bool keep = i < kQueries && !(score < threshold) && !(x2r <= x1r || y2r <= y1r);
    const unsigned long long m = __ballot(keep);
    if (lane == 0) wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int pos = __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += wave_cnt[w];
    if (keep) {
        const float sx = img_w / 640.0f, sy = img_h / 640.0f;
        float* d = dets + (int64_t)pos * kLogitLen;
        d[0] = fmaxf(x1r * sx, 0.0f);
        d[1] = fmaxf(y1r * sy, 0.0f);
        d[2] = fminf(x2r * sx, img_w);
        d[3] = fminf(y2r * sy, img_h);
        d[4] = score;
        // `as usize` saturates (negative / NaN -> 0), then .min(num_classes - 1)
        const float cf = q[5];
        int cid = cf > 0.0f ? (cf >= 2147483520.0f ? 2147483647 : (int)cf) : 0;
        cid = cid < num_classes - 1 ? cid : num_classes - 1;
        d[5] = (float)cid;
        for (int j = 0; j < kMaskDim; ++j) d[6 + j] = q[6 + j];
    }
    const int total = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3] + wave_cnt[4];
    if (i == 0) *count = total;
    // rows [total, 300): zeros (upstream returns a Vec of `total` detections; a fixed-width buffer that travels between ranks must not
    // carry whatever the allocation held).  The kept rows lie below `total`, written by other threads: disjoint.
    for (int r = total + i; r < kQueries; r += blockDim.x)
        for (int j = 0; j < kLogitLen; ++j) dets[(int64_t)r * kLogitLen + j] = 0.0f;
}

// image.rs:213-262: one thread per image pixel; a pixel is set when ANY kept detection covers it (order-free)
__global__ void yolo_mask_kernel(const float* __restrict__ dets_all, const int32_t* __restrict__ count_all, const float* __restrict__ feat_all,
                                 int mask_h, int mask_w, int img_w, int img_h, uint8_t* __restrict__ mask_all) {
    const int total = img_w * img_h, n = count_all[blockIdx.y];   // blockIdx.y = the image
    const float* dets = dets_all + (int64_t)blockIdx.y * kQueries * kLogitLen;
    const float* feat = feat_all + (int64_t)blockIdx.y * kMaskDim * mask_h * mask_w;
    uint8_t* mask_img = mask_all + (int64_t)blockIdx.y * total;
    const float scale_x = (float)mask_w / (float)img_w, scale_y = (float)mask_h / (float)img_h;
    const int plane = mask_h * mask_w;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int iy = i / img_w, ix = i - iy * img_w;
        int mx = (int)floorf(((float)ix + 0.5f) * scale_x), my = (int)floorf(((float)iy + 0.5f) * scale_y);
        mx = mx < mask_w - 1 ? mx : mask_w - 1;
        my = my < mask_h - 1 ? my : mask_h - 1;
        const float* f = feat + my * mask_w + mx;
        uint8_t v = 0;
        for (int d = 0; d < n; ++d) {
            const float* det = dets + (int64_t)d * kLogitLen;
            const bool in_bbox = (float)ix >= det[0] && (float)ix <= det[2] && (float)iy >= det[1] && (float)iy <= det[3];
            if (!in_bbox) continue;
            float sum = 0.0f;
            for (int c = 0; c < kMaskDim; ++c) sum = sum + det[6 + c] * f[(int64_t)c * plane];  // separate multiply and add
            const float e = (float)exp((double)(-sum));  // libm expf upstream: correctly rounded, as this is
            const float mv = 1.0f / (1.0f + e);
            if (mv > 0.5f && mv * det[4] > 0.5f) v = 255;
        }
        mask_img[i] = v;
    }
}

}  // namespace

extern "C" {

int lele_hip_wav_to_f32(LeleCtx* ctx, const LeleTensor* bytes, int32_t bits_per_sample, int32_t num_channels, LeleBuf* out,
                        int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && bytes && out, "wav_to_f32: NULL argument");
    LELE_REQUIRE(bytes->dtype == LELE_U8 || bytes->dtype == LELE_I8, "wav_to_f32: the PCM payload must be a byte tensor");
    LELE_REQUIRE(bits_per_sample == 16 || bits_per_sample == 8, "Unsupported bits per sample: %d", bits_per_sample);  // audio.rs:64
    LELE_REQUIRE(num_channels == 1 || num_channels == 2, "wav_to_f32: %d channels (the reference handles mono and stereo)",
                 num_channels);
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t nbytes = numel(bytes);
    const int64_t samples = bits_per_sample == 16 ? nbytes / 2 : nbytes;  // chunks_exact(2): a trailing odd byte is dropped
    // stereo: samples.chunks(2) would emit a last 1-element chunk and index ch[1] out of bounds -> require whole frames
    LELE_REQUIRE(samples % num_channels == 0, "wav_to_f32: sample count %lld is not a multiple of %d channels",
                 (long long)samples, num_channels);
    const int64_t frames = samples / num_channels;
    LELE_TRY(ctx->arena_reset());
    const void* db = nullptr;
    LELE_TRY(ctx->dev_ptr(bytes, &db));
    LELE_TRY(out->reserve((size_t)frames * 4));
    if (frames) {
        const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>((frames + 255) / 256, 8192));
        hipLaunchKernelGGL(wav_to_f32_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const uint8_t*)db, frames,
                           bits_per_sample, num_channels, (float*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape(out_shape, out_rank, {frames});
}

int lele_hip_argmax_last(LeleCtx* ctx, const LeleTensor* x, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "argmax_last: NULL argument");
    LELE_REQUIRE(x->dtype == LELE_F32 && x->rank >= 1, "argmax_last: f32 tensor of rank >= 1 required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t v = x->shape[x->rank - 1];
    int64_t rows = 1;
    for (int i = 0; i + 1 < x->rank; ++i) rows *= x->shape[i];
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)std::max<int64_t>(rows, 1) * 4));
    if (rows) {
        hipLaunchKernelGGL(argmax_last_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->stream, (const float*)dx, rows, v,
                           (int32_t*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank - 1));
}

int lele_hip_token_filter(LeleCtx* ctx, const LeleTensor* ids, const LeleTensor* skip, LeleBuf* out_ids, LeleBuf* out_counts,
                          int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && ids && skip && out_ids && out_counts, "token_filter: NULL argument");
    LELE_REQUIRE(ids->dtype == LELE_I32 && ids->rank >= 1, "token_filter: i32 ids of rank >= 1 required");
    LELE_REQUIRE((skip->dtype == LELE_U8 || skip->dtype == LELE_I8) && skip->rank == 1, "token_filter: skip must be a byte vector [V]");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t t = ids->shape[ids->rank - 1];
    int64_t rows = 1;
    for (int i = 0; i + 1 < ids->rank; ++i) rows *= ids->shape[i];
    LELE_TRY(ctx->arena_reset());
    const void *di = nullptr, *ds = nullptr;
    LELE_TRY(ctx->dev_ptr(ids, &di));
    LELE_TRY(ctx->dev_ptr(skip, &ds));
    LELE_TRY(out_ids->reserve((size_t)std::max<int64_t>(rows * t, 1) * 4));
    LELE_TRY(out_counts->reserve((size_t)std::max<int64_t>(rows, 1) * 4));
    if (rows) {
        hipLaunchKernelGGL(token_filter_kernel, dim3((unsigned)rows), dim3(256), 0, ctx->stream, (const int32_t*)di, t,
                           (const uint8_t*)ds, skip->shape[0], (int32_t*)out_ids->data, (int32_t*)out_counts->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(ids->shape, ids->shape + ids->rank));
}

int lele_hip_image_preprocess(LeleCtx* ctx, const LeleTensor* rgb, int32_t target, LeleBuf* out, int64_t* out_shape,
                              int32_t* out_rank) {
    LELE_REQUIRE(ctx && rgb && out, "image_preprocess: NULL argument");
    LELE_REQUIRE(rgb->dtype == LELE_U8 && rgb->rank == 3 && rgb->shape[2] == 3, "image_preprocess: u8 [H, W, 3] image required");
    LELE_REQUIRE(rgb->shape[0] > 0 && rgb->shape[1] > 0 && target > 0 && target <= 16384, "image_preprocess: empty image or bad target");
    LELE_REQUIRE(rgb->shape[0] < (1 << 24) && rgb->shape[1] < (1 << 24), "image_preprocess: image side exceeds 2^24");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(ctx->arena_reset());
    const void* dr = nullptr;
    LELE_TRY(ctx->dev_ptr(rgb, &dr));
    LELE_TRY(out->reserve((size_t)3 * target * target * 4));
    const int total = target * target;
    hipLaunchKernelGGL(image_preprocess_kernel, dim3((total + 255) / 256), dim3(256), 0, ctx->stream, (const uint8_t*)dr,
                       (int)rgb->shape[0], (int)rgb->shape[1], target, (float*)out->data);
    LELE_HIP_CHECK(hipGetLastError());
    return set_shape(out_shape, out_rank, {1, 3, target, target});
}

int lele_hip_yolo_seg_postprocess(LeleCtx* ctx, const LeleTensor* logits, const LeleTensor* mask_features, int32_t img_width,
                                  int32_t img_height, float threshold, int32_t num_classes, LeleBuf* out_dets,
                                  LeleBuf* out_count, LeleBuf* out_mask) {
    LELE_REQUIRE(ctx && logits && mask_features && out_dets && out_count && out_mask, "yolo_seg_postprocess: NULL argument");
    const int64_t per = (int64_t)kQueries * kLogitLen;
    LELE_REQUIRE(logits->dtype == LELE_F32 && numel(logits) > 0 && numel(logits) % per == 0,
                 "yolo_seg_postprocess: logits must hold N x 300 x 38 f32 values");
    const int64_t images = numel(logits) / per;   // upstream: one image a call (image.rs:127); a batch is the same routine per image
    LELE_REQUIRE(mask_features->dtype == LELE_F32, "yolo_seg_postprocess: f32 mask features required");
    LELE_REQUIRE(img_width > 0 && img_height > 0 && (int64_t)img_width * img_height < (int64_t(1) << 31) && num_classes > 0,
                 "yolo_seg_postprocess: bad image size or class count");
    LELE_REQUIRE(numel(mask_features) % (images * kMaskDim) == 0 && images < 65536, "yolo_seg_postprocess: mask features do not match %lld images",
                 (long long)images);
    const int64_t mask_hw = numel(mask_features) / (images * kMaskDim);      // image.rs:142-145
    const int mask_h = (int)sqrtf((float)mask_hw), mask_w = mask_h;
    LELE_REQUIRE(mask_h > 0, "yolo_seg_postprocess: empty mask features");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(ctx->arena_reset());
    const void *dl = nullptr, *df = nullptr;
    LELE_TRY(ctx->dev_ptr(logits, &dl));
    LELE_TRY(ctx->dev_ptr(mask_features, &df));
    LELE_TRY(out_dets->reserve((size_t)(images * per) * 4));
    LELE_TRY(out_count->reserve((size_t)images * 4));
    LELE_TRY(out_mask->reserve((size_t)images * img_width * img_height));
    hipLaunchKernelGGL(yolo_dets_kernel, dim3((unsigned)images), dim3(320), 0, ctx->stream, (const float*)dl, (float)img_width, (float)img_height,
                       threshold, num_classes, (float*)out_dets->data, (int32_t*)out_count->data);
    const int total = img_width * img_height;
    hipLaunchKernelGGL(yolo_mask_kernel, dim3(std::min((total + 255) / 256, 16384), (unsigned)images), dim3(256), 0, ctx->stream,
                       (const float*)out_dets->data, (const int32_t*)out_count->data, (const float*)df, mask_h, mask_w, img_width,
                       img_height, (uint8_t*)out_mask->data);
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // extern "C"
