// rnn.hip -- LSTM / GRU forward (batch 1, one direction) on gfx950.
//
//   lele_hip_lstm <- /root/reference/src/kernels/rnn.rs:67-231 (gate math 15-65)
//   lele_hip_gru  <- /root/reference/src/kernels/rnn.rs:246-357 (gate fusion 359-432)
//
// The reference runs, per time step, two faer GEMVs (W x_t, R h_{t-1}) and an AVX2 gate pass.  The recurrence is
// inherently serial in t, so the device splits it differently:
//   1. W x_t does not depend on the recurrence -> ONE MFMA GEMM for all T steps ([T, I] x [I, G] -> WX[T, G]).
//   2. R is transposed once ([H, G]) so that lane g of the recurrent GEMV reads R^T[k][g]: coalesced across lanes,
//      and the k loop of one lane is a plain FMA chain with independent loads (no cross-lane reduction per row).
//   3. One persistent 1024-thread workgroup walks t = 0..T-1 with h, c and the per-step gate pre-activations in
//      LDS; R^T (G*H*4 bytes, 256 KB for the VAD-sized LSTM) stays L2-resident across steps.
// Gate math follows the x86 code exactly: the same association of the adds, polynomial sigmoid/tanh for hidden
// indices k < (H & ~7), libm for the tail, fma for c_t / h_t.  The GRU always evaluates the linear_before_reset = 1
// form, as gru_gate_fusion_avx2 does regardless of its flag (rnn.rs:366, 393-407); the flag is accepted and ignored.
#include "common.h"
#include "gemm_core.h"
#include "simd_math.h"

#include <math.h>

using namespace lele;

namespace {

constexpr int kRnnThreads = 1024;

__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int r = r0 + j, c = c0 + threadIdx.x;
        tile[j][threadIdx.x] = (r < rows && c < cols) ? src[(int64_t)r * cols + c] : 0.0f;
    }
    __syncthreads();
    for (int j = threadIdx.y; j < 32; j += blockDim.y) {
        const int c = c0 + j, r = r0 + threadIdx.x;
        if (r < rows && c < cols) dst[(int64_t)c * rows + r] = tile[threadIdx.x][j];
    }
}

__device__ __forceinline__ float sigmoid_tail(float x) { return 1.0f / (1.0f + expf(-x)); }  // activations.rs sigmoid

// MODE 0 = LSTM (NG = 4 gates: i, o, f, c), MODE 1 = GRU (NG = 3: z, r, h)
template <int MODE>
__global__ __launch_bounds__(kRnnThreads) void rnn_kernel(const float* __restrict__ wx /*[T,G]*/,
                                                          const float* __restrict__ rt /*[H,G]*/,
                                                          const float* __restrict__ bias /*[2G] or null*/,
                                                          const float* __restrict__ h0, const float* __restrict__ c0,
                                                          float* __restrict__ y, float* __restrict__ hout,
                                                          float* __restrict__ cout, int T, int H, int S) {
    constexpr int NG = MODE == 0 ? 4 : 3;
    const int G = NG * H;
    extern __shared__ float lds[];
    float* h = lds;           // [H]
    float* c = lds + H;       // [H] (LSTM only)
    float* part = c + H;      // [S][G] partial recurrent sums
    const int tid = threadIdx.x;
    for (int k = tid; k < H; k += kRnnThreads) {
        h[k] = h0 ? h0[k] : 0.0f;
        c[k] = (MODE == 0 && c0) ? c0[k] : 0.0f;
    }
    __syncthreads();
    const int body = H & ~7;
    for (int t = 0; t < T; ++t) {
        // recurrent GEMV: thread (s, g) sums k in [s*H/S, (s+1)*H/S)
        for (int idx = tid; idx < S * G; idx += kRnnThreads) {
            const int s = idx / G, g = idx - s * G;
            const int k0 = (int)((int64_t)s * H / S), k1 = (int)((int64_t)(s + 1) * H / S);
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                a0 = fmaf_(rt[(int64_t)(k + 0) * G + g], h[k + 0], a0);
                a1 = fmaf_(rt[(int64_t)(k + 1) * G + g], h[k + 1], a1);
                a2 = fmaf_(rt[(int64_t)(k + 2) * G + g], h[k + 2], a2);
                a3 = fmaf_(rt[(int64_t)(k + 3) * G + g], h[k + 3], a3);
            }
            for (; k < k1; ++k) a0 = fmaf_(rt[(int64_t)k * G + g], h[k], a0);
            part[idx] = (a0 + a1) + (a2 + a3);
        }
        __syncthreads();
        const float* wxt = wx + (int64_t)t * G;
        for (int k = tid; k < H; k += kRnnThreads) {
            float rc[NG], wc[NG], bw[NG], br[NG];
#pragma unroll
            for (int q = 0; q < NG; ++q) {
                float acc = part[q * H + k];
                for (int s = 1; s < S; ++s) acc += part[s * G + q * H + k];
                rc[q] = acc;
                wc[q] = wxt[q * H + k];
                bw[q] = bias ? bias[q * H + k] : 0.0f;
                br[q] = bias ? bias[G + q * H + k] : 0.0f;
            }
            const bool poly = k < body;
            float ht;
            if (MODE == 0) {
                float gate[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) gate[q] = ((wc[q] + rc[q]) + bw[q]) + br[q];  // rnn.rs:152-154
                float ct;
                if (poly) {  // lstm_gates_avx2, rnn.rs:26-63
                    const float ig = sigmoid_poly(gate[0]), og = sigmoid_poly(gate[1]), fg = sigmoid_poly(gate[2]);
                    const float cg = tanh_poly(gate[3]);
                    ct = fmaf_(fg, c[k], ig * cg);
                    ht = og * tanh_poly(ct);
                } else {  // scalar tail, rnn.rs:52-63
                    const float ig = sigmoid_tail(gate[0]), og = sigmoid_tail(gate[1]), fg = sigmoid_tail(gate[2]);
                    const float cg = tanhf(gate[3]);
                    ct = fg * c[k] + ig * cg;
                    ht = og * tanhf(ct);
                }
                c[k] = ct;
            } else {
                if (poly) {  // gru_gate_fusion_avx2, rnn.rs:373-416
                    const float z = sigmoid_poly((wc[0] + rc[0]) + (bw[0] + br[0]));
                    const float rg = sigmoid_poly((wc[1] + rc[1]) + (bw[1] + br[1]));
                    const float hg = tanh_poly((wc[2] + bw[2]) + rg * (rc[2] + br[2]));
                    ht = fmaf_(1.0f - z, hg, z * h[k]);
                } else {  // rnn.rs:418-431
                    const float z = sigmoid_tail(((wc[0] + rc[0]) + bw[0]) + br[0]);
                    const float rg = sigmoid_tail(((wc[1] + rc[1]) + bw[1]) + br[1]);
                    const float hg = tanhf((wc[2] + bw[2]) + rg * (rc[2] + br[2]));
                    ht = (1.0f - z) * hg + z * h[k];
                }
            }
            h[k] = ht;  // only this thread reads h[k] in the gate phase; the next GEMV starts after the barrier
            y[(int64_t)t * H + k] = ht;
        }
        __syncthreads();
    }
    for (int k = tid; k < H; k += kRnnThreads) {
        hout[k] = h[k];
        if (MODE == 0) cout[k] = c[k];
    }
}

template <int MODE>
int run_rnn(LeleCtx* ctx, const char* name, const LeleTensor* x, const LeleTensor* w, const LeleTensor* r,
            const LeleTensor* bias, const LeleTensor* h0, const LeleTensor* c0, LeleBuf* y, LeleBuf* hn, LeleBuf* cn,
            int64_t* y_shape, int32_t* y_rank) {
    constexpr int NG = MODE == 0 ? 4 : 3;
    LELE_REQUIRE(ctx && x && w && r && y && hn && (MODE == 1 || cn), "%s: NULL argument", name);
    LELE_REQUIRE(x->rank == 3 && w->rank == 3 && r->rank == 3, "%s: expected X [T,B,I], W [D,%dH,I], R [D,%dH,H]", name,
                 NG, NG);
    LELE_REQUIRE(x->dtype == LELE_F32 && w->dtype == LELE_F32 && r->dtype == LELE_F32, "%s: f32 tensors required", name);
    LELE_REQUIRE(w->shape[0] == 1, "%s: Only num_directions=1 supported", name);  // rnn.rs:86, 266
    LELE_REQUIRE(x->shape[1] == 1, "%s: Only batch_size=1 supported", name);     // rnn.rs:89, 269
    const int64_t T = x->shape[0], I = x->shape[2], H = w->shape[1] / NG, G = NG * H;
    LELE_REQUIRE(w->shape[1] == G && w->shape[2] == I, "%s: W shape mismatch", name);
    LELE_REQUIRE(r->shape[0] == 1 && r->shape[1] == G && r->shape[2] == H, "%s: R shape mismatch", name);
    if (bias) LELE_REQUIRE(numel(bias) == 2 * G, "%s: bias must hold %lld values", name, (long long)(2 * G));
    if (h0) LELE_REQUIRE(numel(h0) == H, "%s: initial_h must hold %lld values", name, (long long)H);
    if (c0) LELE_REQUIRE(numel(c0) == H, "%s: initial_c must hold %lld values", name, (long long)H);
    LELE_REQUIRE(H >= 1, "%s: hidden_size must be positive", name);
    int S = (int)std::max<int64_t>(1, kRnnThreads / G);
    S = (int)std::min<int64_t>(S, H);
    const size_t lds_bytes = (size_t)(2 * H + (int64_t)S * G) * 4;
    LELE_REQUIRE(lds_bytes <= 160 * 1024, "%s: hidden_size %lld exceeds the LDS-resident state limit", name, (long long)H);
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dw = nullptr, *dr = nullptr, *db = nullptr, *dh0 = nullptr, *dc0 = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(w, &dw));
    LELE_TRY(ctx->dev_ptr(r, &dr));
    if (bias) LELE_TRY(ctx->dev_ptr(bias, &db));
    if (h0) LELE_TRY(ctx->dev_ptr(h0, &dh0));
    if (c0) LELE_TRY(ctx->dev_ptr(c0, &dc0));
    void *wx = nullptr, *rt = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)std::max<int64_t>(1, T * G) * 4, &wx));
    // R transposed: once per weight when the caller declared it immutable (a streaming model calls this for every chunk)
    const bool r_cacheable = r->mem == LELE_MEM_WEIGHT;
    const auto r_key = std::make_tuple((const void*)r->data, (size_t)G * H * 4, 601);
    bool have_rt = false;
    if (r_cacheable) {
        auto it = ctx->weights.find(r_key);
        if (it != ctx->weights.end()) {
            rt = it->second;
            have_rt = true;
        } else {
            LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
            LELE_HIP_CHECK(hipMalloc(&rt, (size_t)G * H * 4));
            ctx->weights[r_key] = rt;
        }
    } else {
        LELE_TRY(ctx->arena_alloc((size_t)G * H * 4, &rt));
    }
    LELE_TRY(y->reserve((size_t)T * H * 4));
    LELE_TRY(hn->reserve((size_t)H * 4));
    if (MODE == 0) LELE_TRY(cn->reserve((size_t)H * 4));
    if (T > 0) {
        gemm::LoadRowK al{(const float*)dx, 0, I, (int)T, (int)I, (int)((((uintptr_t)dx & 15) == 0) && I % 4 == 0)};
        gemm::LoadRowK bl{(const float*)dw, 0, I, (int)G, (int)I, (int)((((uintptr_t)dw & 15) == 0) && I % 4 == 0)};
        gemm::EpiAffine epi{(float*)wx, 0, (int)T, (int)G, 1.0f, 0.0f, nullptr, gemm::C_NONE, 1};
        gemm::launch(ctx->stream, al, bl, epi, (int)T, (int)G, (int)I, 1, ctx->num_cus);
    }
    if (!have_rt)
        hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)((H + 31) / 32), (unsigned)((G + 31) / 32)), dim3(32, 8), 0,
                           ctx->stream, (const float*)dr, (float*)rt, (int)G, (int)H);
    LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(&rnn_kernel<MODE>), 160 * 1024));
    hipLaunchKernelGGL(rnn_kernel<MODE>, dim3(1), dim3(kRnnThreads), lds_bytes, ctx->stream, (const float*)wx,
                       (const float*)rt, (const float*)db, (const float*)dh0, (const float*)dc0, (float*)y->data,
                       (float*)hn->data, MODE == 0 ? (float*)cn->data : nullptr, (int)T, (int)H, S);
    LELE_HIP_CHECK(hipGetLastError());
    return set_shape(y_shape, y_rank, {T, 1, 1, H});  // rnn.rs:223, 352; h / c are [1, 1, H]
}

}  // namespace

extern "C" {

int lele_hip_lstm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* r, const LeleTensor* bias,
                  const LeleTensor* sequence_lens, const LeleTensor* initial_h, const LeleTensor* initial_c, LeleBuf* out_y,
                  LeleBuf* out_h, LeleBuf* out_c, int64_t* y_shape, int32_t* y_rank) {
    (void)sequence_lens;  // ignored by the reference as well (rnn.rs:72)
    return run_rnn<0>(ctx, "LSTM", x, w, r, bias, initial_h, initial_c, out_y, out_h, out_c, y_shape, y_rank);
}

int lele_hip_gru(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* r, const LeleTensor* bias,
                 const LeleTensor* initial_h, int linear_before_reset, LeleBuf* out_y, LeleBuf* out_h, int64_t* y_shape,
                 int32_t* y_rank) {
    (void)linear_before_reset;  // the x86 gate fusion evaluates the =1 form for either value (rnn.rs:366)
    return run_rnn<1>(ctx, "GRU", x, w, r, bias, initial_h, nullptr, out_y, out_h, nullptr, y_shape, y_rank);
}

}  // extern "C"
