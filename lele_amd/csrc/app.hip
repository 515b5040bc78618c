// app.hip -- the application-side pre/post-processing steps that bracket a lele model run,
// moved to the device so that (a) 16-bit PCM crosses PCIe instead of f32 and (b) only token ids leave the GPU.
//
//   lele_hip_wav_to_f32    <- /root/reference/examples/sensevoice/src/audio.rs:52-73   (s16 / u8 -> f32, stereo -> mono)
//   lele_hip_argmax_last   <- /root/reference/examples/sensevoice/src/tokenizer.rs:50-61 (per-frame arg-max of the logits;
//                             Iterator::max_by keeps the LAST of equal maxima)
// Both are exact: one IEEE operation per sample / pure comparisons.
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

}  // extern "C"
