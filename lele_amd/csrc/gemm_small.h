// gemm_small.h -- latency-optimised GEMMs for problems that are too small to fill the chip with 64x64 / 128x128 tiles
// (SenseVoice at M = 504: attention products, per-layer linears).
//
// One 256-thread workgroup per 32x32 output tile; its four waves split K four ways.  Every wave feeds its MFMAs
// straight from global memory (the operands are L2-resident at these sizes): no LDS staging, no barrier in the K loop,
// all loads of a K step issued back to back.  The four partial tiles meet once in LDS and are added in a fixed order
// (p0+p1)+(p2+p3) -- deterministic; for f32 a re-association of the exact-product sum (inside the 1e-4 bar), for i8 exact.
//   f32: v_mfma_f32_32x32x2_f32, lane (l31, hv) owns k = 16c + 8hv + s of chunk c   (same operand mapping as gemm_core.h)
//   i8 : v_mfma_i32_32x32x32_i8, lane (l31, hv) owns the 16 bytes at k = 32s + 16hv of step s
#pragma once
#include "gemm_core.h"

namespace gemm {

typedef int sm_v4i __attribute__((ext_vector_type(4)));
typedef int sm_v16i __attribute__((ext_vector_type(16)));

// A: element(row, k) = a[batch*bsA + row*lda + k]  (k contiguous).
// B: B_KCONTIG ? b[batch*bsB + col*ldb + k] : b[batch*bsB + k*ldb + col].
template <bool B_KCONTIG, class EPI>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(const float* __restrict__ a, int64_t bsA, int64_t lda,
                                                             const float* __restrict__ b, int64_t bsB, int64_t ldb,
                                                             EPI epi, int M, int N, int K, int vecA, int vecB) {
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hv = lane >> 5, l31 = lane & 31;
    const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32, batch = blockIdx.z;
    const int row = m0 + l31, col = n0 + l31;
    const bool rin = row < M, cin = col < N;
    const float* ap = a + (int64_t)batch * bsA + (int64_t)(rin ? row : M - 1) * lda;
    const float* bp = b + (int64_t)batch * bsB + (B_KCONTIG ? (int64_t)(cin ? col : N - 1) * ldb : (int64_t)(cin ? col : N - 1));
    const int nchunk = (K + 15) / 16, per = (nchunk + 3) / 4;
    const int c0 = wave * per, c1 = c0 + per < nchunk ? c0 + per : nchunk;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

    auto load8 = [&](const float* p, int64_t stride_k, int k0, bool contig, int vec, bool in, float (&f)[8]) {
        // 8 consecutive k starting at k0 for this lane's row / column; out-of-range -> 0 (clamped, unconditional loads)
        if (contig && vec && k0 + 8 <= K) {
            const float4 v0 = *reinterpret_cast<const float4*>(p + k0), v1 = *reinterpret_cast<const float4*>(p + k0 + 4);
            f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w;
            f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = k0 + e < K ? k0 + e : K - 1;
                f[e] = p[(int64_t)k * stride_k];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (k0 + e >= K) f[e] = 0.0f;
        }
        if (!in) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = 0.0f;
        }
    };

    for (int c = c0; c < c1; c += 2) {  // two 16-k chunks per trip: 32 loads in flight per lane
        float fa[2][8], fb[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int k0 = (c + u) * 16 + 8 * hv;
            const bool live = c + u < c1;
            load8(ap, 1, live ? k0 : 0, true, vecA, rin && live, fa[u]);
            load8(bp, B_KCONTIG ? 1 : ldb, live ? k0 : 0, B_KCONTIG, vecB, cin && live, fb[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[u][s], fb[u][s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // wave w finishes accumulator registers 4w..4w+3 of the tile: row = (r&3) + 8*(r>>2) + 4*hv, col = l31
    const int colc = cin ? col : N - 1;
    float pre[4], tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        const int orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * hv;
        pre[q] = epi.load(batch, orow < M ? orow : M - 1, colc);
        tot[q] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        epi.store(batch, m0 + (r & 3) + 8 * (r >> 2) + 4 * hv, col, tot[q], pre[q]);
    }
}

// i8: A' [rows][kp], B' [n][kp], both k-contiguous, kp a multiple of 16.  EPI = quant.hip's IgemmEpi.
template <class EPI>
__global__ __launch_bounds__(256) void igemm_small_kernel(const int8_t* __restrict__ a, const int8_t* __restrict__ b,
                                                          int64_t rows, int n, int kp, int64_t b_batch_stride,
                                                          int m_per_batch, EPI epi) {
    __shared__ int red[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hv = lane >> 5, l31 = lane & 31;
    const int64_t m0 = (int64_t)blockIdx.y * 32;
    const int n0 = blockIdx.x * 32;
    const int64_t row = m0 + l31;
    const int col = n0 + l31;
    const bool rin = row < rows, cin = col < n;
    const int8_t* ap = a + (rin ? row : rows - 1) * kp;
    const int8_t* bp = b + (b_batch_stride ? (m0 / m_per_batch) * b_batch_stride : 0) + (int64_t)(cin ? col : n - 1) * kp;
    const int nstep = (kp + 31) / 32, per = (nstep + 3) / 4;
    const int s0 = wave * per, s1 = s0 + per < nstep ? s0 + per : nstep;
    sm_v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    const sm_v4i zero4 = {0, 0, 0, 0};
    for (int s = s0; s < s1; s += 4) {  // four MFMA steps (128 bytes of K) per trip: 8 x 16-byte loads in flight
        sm_v4i fa[4], fb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = (s + u) * 32 + 16 * hv;
            const bool live = s + u < s1 && k < kp;
            const int kc = live ? k : 0;
            const sm_v4i va = *reinterpret_cast<const sm_v4i*>(ap + kc), vb = *reinterpret_cast<const sm_v4i*>(bp + kc);
            fa[u] = live && rin ? va : zero4;
            fb[u] = live && cin ? vb : zero4;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u], fb[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    const typename EPI::ColCtx cc = epi.col_ctx(col);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * hv;
        const typename EPI::RowCtx rc = epi.row_ctx(orow < rows ? orow : rows - 1);  // clamped: load unconditionally
        const int tot = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
        if (orow < rows && cin) epi.store(rc, cc, col, tot);
    }
}

}  // namespace gemm
