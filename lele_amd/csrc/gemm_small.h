// gemm_small.h -- latency-optimised GEMMs for problems that are too small to fill the chip with 64x64 / 128x128 tiles
// (SenseVoice at M = 504: attention products, per-layer linears).
//
// One 256-thread workgroup per 32x32 output tile; its four waves split K four ways.  Every wave feeds its MFMAs
// straight from global memory (the operands are L2-resident at these sizes): no LDS staging, no barrier in the K loop,
// all loads of a K step issued back to back.  The four partial tiles meet once in LDS and are added in a fixed order
// (p0+p1)+(p2+p3) -- deterministic; for f32 a re-association of the exact-product sum (inside the 1e-4 bar), for i8 exact.
//   f32: gemm_f32_small_kernel lives in gemm_core.h (it takes the same loader / epilogue functors as the tiled kernel)
//   i8 : v_mfma_i32_32x32x32_i8, lane (l31, hv) owns the 16 bytes at k = 32s + 16hv of step s
#pragma once
#include "gemm_core.h"

namespace gemm {

typedef int sm_v4i __attribute__((ext_vector_type(4)));
typedef int sm_v16i __attribute__((ext_vector_type(16)));

// i8: A' [rows][kp], B' [n][kp], both k-contiguous, kp a multiple of 16.  EPI = quant.hip's IgemmEpi.
// STEPS = MFMA steps (32 bytes of K each) a wave keeps in flight per trip: 4 for K <= 512 (few registers: three workgroups per CU
// fit, which the 768 tiles of a 504 x 1536 result need), 16 for longer K (one trip up to K = 2048)
template <class EPI, int STEPS>
__global__ __launch_bounds__(256) void igemm_small_kernel(const int8_t* __restrict__ a, const int8_t* __restrict__ b,
                                                          int64_t rows, int n, int kp, int64_t b_batch_stride,
                                                          int m_per_batch, EPI epi) {
    __shared__ int red[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hv = lane >> 5, l31 = lane & 31;
    unsigned tx, ty, tz;
    tile_coords(tx, ty, tz);  // XCD-aware order (gemm_core.h)
    const int64_t m0 = (int64_t)ty * 32;
    const int n0 = tx * 32;
    const int64_t row = m0 + l31;
    const int col = n0 + l31;
    const bool rin = row < rows, cin = col < n;
    const int8_t* ap = a + (rin ? row : rows - 1) * kp;
    const int8_t* bp = b + (b_batch_stride ? (m0 / m_per_batch) * b_batch_stride : 0) + (int64_t)(cin ? col : n - 1) * kp;
    const int nstep = (kp + 31) / 32, per = (nstep + 3) / 4;
    const int s0 = wave * per, s1 = s0 + per < nstep ? s0 + per : nstep;
    // Everything the epilogue needs from memory is requested FIRST -- column terms, the row terms and residual operands of the four
    // rows this wave finishes -- so that it travels with the operand fragments: this kernel is one chain of memory round trips
    // (launch, fragments, epilogue terms, store), and every trip taken off the chain is ~1 us of a ~8 us kernel
    const typename EPI::ColCtx cc = epi.col_ctx(col);
    typename EPI::RowCtx rcs[4];
    float r1[4] = {0.f, 0.f, 0.f, 0.f}, r2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * hv;
        const int64_t oc = orow < rows ? orow : rows - 1;  // clamped: load unconditionally
        rcs[q] = epi.row_ctx(oc);
        if (epi.res1) {
            const int64_t at = oc * (int64_t)n + (cin ? col : n - 1);
            r1[q] = epi.res1[at];
            r2[q] = epi.res2 ? epi.res2[at] : 0.0f;
        }
    }
    sm_v16i acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0;
    const sm_v4i zero4 = {0, 0, 0, 0};
    // a trip's fragments are ALL in flight before its first MFMA (STEPS = 16: 512 bytes of K per wave = one trip up to K = 2048)
    for (int sb = s0; sb < s1; sb += STEPS) {
        sm_v4i fa[STEPS], fb[STEPS];
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            if (u < 4 || sb + u < s1) {  // uniform; the first four are unconditional (clamped) so that short K stays branch-free
                const int k = (sb + u) * 32 + 16 * hv;
                const bool live = sb + u < s1 && k < kp;
                const int kc = live ? k : 0;
                const sm_v4i va = *reinterpret_cast<const sm_v4i*>(ap + kc), vb = *reinterpret_cast<const sm_v4i*>(bp + kc);
                fa[u] = live && rin ? va : zero4;
                fb[u] = live && cin ? vb : zero4;
            }
        }
#pragma unroll
        for (int u = 0; u < STEPS; ++u)
            if (u < 4 || sb + u < s1) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[u], fb[u], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    float smn = 3.40282347e+38f, smx = -3.40282347e+38f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        const int64_t orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * hv;
        const typename EPI::RowCtx& rc = rcs[q];
        const int tot = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
        if (orow < rows && cin) {
            const float v = epi.res1 ? epi.value_res(rc, cc, tot, r1[q], r2[q]) : epi.value(rc, cc, tot);
            rc.orow[col] = v;
            smn = v < smn ? v : smn;
            smx = v > smx ? v : smx;
        }
    }
    if (epi.blockstat)  // uniform: one {min, max} pair per workgroup for the dynamic quantisation that reads this result next
        block_minmax_store(smn, smx, epi.blockstat + 2 * (size_t)(blockIdx.x + gridDim.x * blockIdx.y));
}

}  // namespace gemm
