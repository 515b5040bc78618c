// igemm_big.h -- the i8 GEMM for COMPUTE-bound shapes (included by quant.hip behind igemm_rs.h).
//
// Same arithmetic as igemm_kernel (exact i32 products on v_mfma_i32_32x32x32_i8, IgemmEpi's f32 epilogue:
// /root/reference/src/kernels/avx/quantization.rs:225-417, 1396-1428), sized for products whose K is long enough that the matrix cores
// are the bound (the 128 x 128 kernel reaches 0.25-0.30 of the i8 peak at 8192 x 4096 x 4096 -- eight waves of a
// 64 x 32 result each spend their time in global -> VGPR -> LDS round trips and two barriers a K step).  Here:
//
//   * a 256 x 256 result a workgroup, FOUR waves (one a SIMD, up to 512 registers each) of 128 x 128: 16 accumulator tiles = 256
//     registers a lane, 64 MFMAs per 128-byte K step against 32 fragment reads (8 KB of LDS a wave);
//   * both operands arrive by direct-to-LDS loads (global_load_lds_dwordx4: no register staging, no ds_write), a whole 128-byte row
//     segment by eight consecutive lanes -- every request is a full cache line -- two 64 KB stages, ONE barrier a K step;
//   * the LDS image of such a load is lane-linear (row pitch 128 bytes: rows two apart would share their banks), so the loader
//     permutes the 16-byte chunks on the GLOBAL side: position p of row r holds chunk p ^ ((r >> 1) & 7); the reader of chunk g asks
//     for position g ^ ((r >> 1) & 7) -- for the lane groups of a ds_read_b128 all positions of a bank half are distinct;
//   * the weights are the MFMA's first operand, so a lane owns one result row and four consecutive columns per register quad:
//     16-byte stores; row / column terms of the epilogue wait in LDS tables filled before the K loop.
//
// Operands: a [rows][kp] and b [n][kp] as for igemm_kernel (q - 128 and w - 128, k contiguous), kp a multiple of 128, n a multiple
// of 4; rows / columns beyond the ends are read from the last valid one and never stored.
#pragma once

namespace {

constexpr int BG_BM = 256, BG_BN = 256, BG_BK = 128;
constexpr int BG_STAGE = (BG_BM + BG_BN) * BG_BK;        // 64 KB: A rows then B rows, 128 bytes each
constexpr int BG_TABLES = 6 * 256 * 4;                   // row terms (ca, rterm, dyn_scale) and column terms (colsum, scale, bias)
constexpr int BG_LDS = 2 * BG_STAGE + BG_TABLES;

__global__ __launch_bounds__(256) void igemm_big_kernel(const int8_t* __restrict__ a, const int8_t* __restrict__ b, int64_t rows, int n, int kp,
                                                        IgemmEpi epi) {
    extern __shared__ __attribute__((aligned(16))) char bg_lds[];
    int* const s_ca = reinterpret_cast<int*>(bg_lds + 2 * BG_STAGE);
    int* const s_rterm = s_ca + 256;
    float* const s_ds = reinterpret_cast<float*>(s_rterm + 256);
    int* const s_colsum = reinterpret_cast<int*>(s_ds + 256);
    float* const s_ws = reinterpret_cast<float*>(s_colsum + 256);
    float* const s_bias = s_ws + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;   // the wave's 128 rows / 128 columns of the workgroup's result
    const int hv = lane >> 5, l31 = lane & 31;
    unsigned tx, ty, tz;
    gemm::tile_coords(tx, ty, tz);             // XCD-aware order (gemm_core.h)
    const int64_t m0 = (int64_t)ty * BG_BM;
    const int n0 = (int)tx * BG_BN;
    {   // epilogue tables (clamped, unconditional loads: their latency hides behind the K loop)
        const int64_t r = m0 + tid < rows ? m0 + tid : rows - 1;
        const IgemmEpi::RowCtx rc = epi.row_ctx(r);
        s_ca[tid] = rc.ca;
        s_rterm[tid] = rc.rterm;
        s_ds[tid] = rc.dyn_scale;
        const IgemmEpi::ColCtx cc = epi.col_ctx(n0 + tid);
        s_colsum[tid] = cc.colsum;
        s_ws[tid] = cc.ws;
        s_bias[tid] = cc.bias;
    }
    // ---- loader: wave w brings rows [64 w, 64 w + 64) of A and of B, eight rows (8 lanes x 16 bytes each) an instruction
    const int rsub = lane >> 3, p = lane & 7;
    const char* asrc[8];
    const char* bsrc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int rl = 64 * wave + 8 * j + rsub;
        const int g = p ^ ((rl >> 1) & 7);
        const int64_t ar = m0 + rl < rows ? m0 + rl : rows - 1;
        const int bc = n0 + rl < n ? n0 + rl : n - 1;
        asrc[j] = reinterpret_cast<const char*>(a) + ar * (int64_t)kp + 16 * g;
        bsrc[j] = reinterpret_cast<const char*>(b) + (int64_t)bc * kp + 16 * g;
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)bg_lds;
    auto issue = [&](int stage, int k0) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)stage * BG_STAGE + (unsigned)(64 * wave) * BG_BK);
#pragma unroll
        for (int j = 0; j < 8; ++j) rs_dma16(asrc[j] + k0, dst + j * 8 * BG_BK);
#pragma unroll
        for (int j = 0; j < 8; ++j) rs_dma16(bsrc[j] + k0, dst + BG_BM * BG_BK + j * 8 * BG_BK);
    };
    // ---- reader: lane (row l31 of a tile, half hv) takes chunk 2 s + hv of its row for k-step s: at position (2 s + hv) ^ swizzle
    const int swz = (l31 >> 1) & 7;
    int roff[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) roff[s] = l31 * BG_BK + (((2 * s + hv) ^ swz) * 16);
    v16i acc[4][4];   // [column tile][row tile]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
    const int nk = kp / BG_BK;
    issue(0, 0);
    rs_wait_vm<0>();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) issue(cur ^ 1, (kt + 1) * BG_BK);
        const char* const abase = bg_lds + cur * BG_STAGE + (128 * wr) * BG_BK;
        const char* const bbase = bg_lds + cur * BG_STAGE + BG_BM * BG_BK + (128 * wc) * BG_BK;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v4i fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v4i*>(abase + i * 32 * BG_BK + roff[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const v4i*>(bbase + j * 32 * BG_BK + roff[s]);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[j][i], 0, 0, 0);
        }
        rs_wait_vm<0>();   // the next stage has landed (this wave's part) ...
        __syncthreads();   // ... everybody's, and nobody reads this stage any more
    }
    // ---- epilogue: lane = result row 128 wr + 32 i + l31, columns 128 wc + 32 j + 8 g + 4 hv + [0, 4)
    const bool two_res = epi.res2 != nullptr;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rl = 128 * wr + 32 * i + l31;
        const int64_t row = m0 + rl;
        const bool rok = row < rows;
        const IgemmEpi::RowCtx rc{s_ca[rl], s_rterm[rl], s_ds[rl], nullptr};
        float* const orow = epi.out + (rok ? row : 0) * (int64_t)n;
        const float* const r1row = epi.res1 ? epi.res1 + (rok ? row : 0) * (int64_t)n : nullptr;
        const float* const r2row = two_res ? epi.res2 + (rok ? row : 0) * (int64_t)n : nullptr;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = 128 * wc + 32 * j + 8 * g + 4 * hv;
                const int col = n0 + cl;
                const bool ok = rok && col < n;   // n % 4 == 0: a group of four columns is whole or absent
                const v4i cs = *reinterpret_cast<const v4i*>(s_colsum + cl);
                const float4 ws = *reinterpret_cast<const float4*>(s_ws + cl);
                const float4 bs = *reinterpret_cast<const float4*>(s_bias + cl);
                float4 o;
                o.x = epi.value24(rc, IgemmEpi::ColCtx{cs[0], ws.x, bs.x}, acc[j][i][4 * g + 0]);
                o.y = epi.value24(rc, IgemmEpi::ColCtx{cs[1], ws.y, bs.y}, acc[j][i][4 * g + 1]);
                o.z = epi.value24(rc, IgemmEpi::ColCtx{cs[2], ws.z, bs.z}, acc[j][i][4 * g + 2]);
                o.w = epi.value24(rc, IgemmEpi::ColCtx{cs[3], ws.w, bs.w}, acc[j][i][4 * g + 3]);
                if (ok) {
                    if (r1row) {
                        const float4 r1 = *reinterpret_cast<const float4*>(r1row + col);
                        o.x = o.x + r1.x, o.y = o.y + r1.y, o.z = o.z + r1.z, o.w = o.w + r1.w;
                        if (r2row) {
                            const float4 r2 = *reinterpret_cast<const float4*>(r2row + col);
                            o.x = o.x + r2.x, o.y = o.y + r2.y, o.z = o.z + r2.z, o.w = o.w + r2.w;
                        }
                    }
                    *reinterpret_cast<float4*>(orow + col) = o;
                }
            }
        }
    }
}

}  // namespace
