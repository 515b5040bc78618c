// igemm_big.h -- the i8 GEMM for COMPUTE-bound shapes (included by quant.hip behind igemm_rs.h).
//
// Same arithmetic as igemm_kernel (exact i32 products on v_mfma_i32_32x32x32_i8, IgemmEpi's f32 epilogue:
// /root/reference/src/kernels/avx/quantization.rs:225-417, 1396-1428), sized for products whose K is long enough that the matrix cores
// are the bound (the 128 x 128 kernel takes 190 us for 8192 x 4096 x 4096 -- eight waves of a 64 x 32 result each
// spend their time in global -> VGPR -> LDS round trips and two barriers a K step).  The shipped form (igemm_big_kernel<4, true>):
//
//   * a 256 x 256 result a workgroup, EIGHT waves (two a SIMD) of 128 x 64: 8 accumulator tiles = 128 registers a lane; while one of
//     a SIMD's two waves requests, reads or waits, the other one's MFMAs keep the matrix core busy;
//   * K in steps of 64 bytes through FOUR 32 KB stages of LDS; the operands go from memory straight into LDS
//     (global_load_lds_dwordx4: no staging registers, no ds_write), two steps requested at a time, ONE raw s_barrier per TWO steps with
//     nothing but the waits for those pieces and for the LDS reads in front of it;
//   * LDS rows are 64 bytes (rows four apart would share their banks), so the 16-byte chunks of a row are permuted: position p of row r
//     holds chunk p ^ ((r >> 2) & 3) -- applied on the GLOBAL side, since the LDS image of a direct load is lane-linear -- and the reader
//     of chunk g asks for position g ^ ((r >> 2) & 3): for the lane groups of a ds_read_b128 all 64 banks are distinct
//     (SQ_LDS_BANK_CONFLICT = 0);
//   * the weights are the MFMA's first operand, so a lane owns one result row and four consecutive columns per register quad:
//     16-byte stores; row / column terms of the epilogue wait in LDS tables filled before the K loop.
//
// How it got there (kernel time for 8192 x 4096 x 4096, rocprofv3; profiles/r06_igemm_big_ladder.json): four waves of 128 x 128 (256
// accumulator registers, one wave a SIMD) with direct-to-LDS loads, 128-byte steps, two stages, drain + __syncthreads: 180-200 us;
// the same through staging registers: 219 (the scheduler sinks the 16 loads of a step to its end, in front of the stores that wait for
// them); loads pinned + fragments double-buffered: 210; four 64-byte stages, early fragment request, raw barrier: 212; one LDS operation
// per MFMA gap (sched_group_barrier): 193; eight waves of 128 x 64: 182; direct-to-LDS loads again: 169 (1.63 POP/s); m0 declared clobbered instead of saved / restored around every piece and two K steps a
// barrier: 169 there, 32768 x 4096 x 2048 301 -> **284-288 us (1.91-1.93 POP/s, 0.48-0.49 of 3944)**.  The ceiling is
// not 3944: tools/mfma_i8_rate.hip measures 4.17 POP/s for this instruction on constant operands and **3.23 on random bytes** (the chip
// clocks down: 16.1 -> 20.8 ns per MFMA and SIMD); of that the kernel reaches 0.50-0.60.  What it is short of (PMC, profiles/
// r06_igemm_big_pmc.txt): the waves spend 27 % of their life in front of the step's barrier / counted waits and 52 % in issue stalls
// of which the matrix core's own occupancy explains 25 points -- the schedule of the guide's 8-phase template (four phases a K step,
// two wave groups staggered by a barrier) is what remains to be built on top of this data path.
//
// Operands: a [rows][kp] and b [n][kp] as for igemm_kernel (q - 128 and w - 128, k contiguous), kp a multiple of 128, n a multiple
// of 4; rows / columns beyond the ends are read from the last valid one and never stored.
#pragma once

namespace {

constexpr int BG_BM = 256, BG_BN = 256, BG_BK = 64, BG_NST = 4;
constexpr int BG_STAGE = (BG_BM + BG_BN) * BG_BK;        // 32 KB: A rows then B rows, 64 bytes each
constexpr int BG_TABLES = 6 * 256 * 4;                   // row terms (ca, rterm, dyn_scale) and column terms (colsum, scale, bias)
constexpr int BG_LDS = BG_NST * BG_STAGE + BG_TABLES;

// WC = waves along the columns: 2 (four waves of 128 x 128, one a SIMD) or 4 (eight waves of 128 x 64, two a SIMD: while one of a SIMD's
// two waves stores, reads or waits, the other one's MFMAs keep the matrix core busy)
// DMA: the operands go from memory straight into LDS (global_load_lds_dwordx4; the 16-byte chunks permuted on the GLOBAL side, since the
// LDS image of such a load is lane-linear): no staging registers, no ds_write -- three K steps in flight, counted vmcnt waits
__device__ __forceinline__ void bg_dma16(const void* base, unsigned voff, unsigned lds_dst) {
    // m0 is declared clobbered instead of being saved and restored around every piece (three scalar instructions a piece less)
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(base), "s"(lds_dst) : "memory", "m0");
}
template <int WC, bool DMA = false>
__global__ __launch_bounds__(128 * WC) void igemm_big_kernel(const int8_t* __restrict__ a, const int8_t* __restrict__ b, int64_t rows, int n, int kp,
                                                            IgemmEpi epi) {
    constexpr int NW = 2 * WC;          // waves
    constexpr int TJ = 256 / WC / 32;   // 32-column tiles a wave
    constexpr int LR = 256 / NW;        // rows of A (and of B) a wave brings per K step
    constexpr int LI = LR / 16;         // ... in this many instructions of 16 rows
    extern __shared__ __attribute__((aligned(16))) char bg_lds[];
    int* const s_ca = reinterpret_cast<int*>(bg_lds + BG_NST * BG_STAGE);
    int* const s_rterm = s_ca + 256;
    float* const s_ds = reinterpret_cast<float*>(s_rterm + 256);
    int* const s_colsum = reinterpret_cast<int*>(s_ds + 256);
    float* const s_ws = reinterpret_cast<float*>(s_colsum + 256);
    float* const s_bias = s_ws + 256;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk_ = kp / BG_BK;
    const int wr = wave / WC, wc = wave % WC;   // the wave's 128 rows / 256 / WC columns of the workgroup's result
    const int hv = lane >> 5, l31 = lane & 31;
    unsigned tx, ty, tz;
    gemm::tile_coords(tx, ty, tz);             // XCD-aware order (gemm_core.h)
    const int64_t m0 = (int64_t)ty * BG_BM;
    const int n0 = (int)tx * BG_BN;
    {   // epilogue tables (clamped, unconditional loads: their latency hides behind the K loop)
        if (tid < 256) {
            const int64_t r = m0 + tid < rows ? m0 + tid : rows - 1;
            const IgemmEpi::RowCtx rc = epi.row_ctx(r);
            s_ca[tid] = rc.ca;
            s_rterm[tid] = rc.rterm;
            s_ds[tid] = rc.dyn_scale;
        } else if (tid < 512) {
            const IgemmEpi::ColCtx cc = epi.col_ctx(n0 + tid - 256);
            s_colsum[tid - 256] = cc.colsum;
            s_ws[tid - 256] = cc.ws;
            s_bias[tid - 256] = cc.bias;
        }
        if (NW == 4) {
            const IgemmEpi::ColCtx cc = epi.col_ctx(n0 + tid);
            s_colsum[tid] = cc.colsum;
            s_ws[tid] = cc.ws;
            s_bias[tid] = cc.bias;
        }
    }
    // ---- loader: wave w brings rows [LR w, LR w + LR) of A and of B, sixteen rows (4 lanes x 16 bytes each) an instruction
    const int rsub = lane >> 2, p = lane & 3;
    unsigned aoff[LI], boff[LI];   // byte offsets from a / b (rows * kp < 2^32: launch_igemm checks)
    int woff[LI];                  // where the chunk goes inside a stage's A (or B) region
#pragma unroll
    for (int j = 0; j < LI; ++j) {
        const int rl = LR * wave + 16 * j + rsub;
        const int64_t ar = m0 + rl < rows ? m0 + rl : rows - 1;
        const int bc = n0 + rl < n ? n0 + rl : n - 1;
        aoff[j] = (unsigned)(ar * (int64_t)kp) + 16u * (unsigned)p;
        boff[j] = (unsigned)((int64_t)bc * kp) + 16u * (unsigned)p;
        woff[j] = rl * BG_BK + ((p ^ ((rl >> 2) & 3)) * 16);
        if (DMA) {   // the permutation on the source side: position p of the row receives chunk p ^ swizzle
            aoff[j] = (unsigned)(ar * (int64_t)kp) + 16u * (unsigned)(p ^ ((rl >> 2) & 3));
            boff[j] = (unsigned)((int64_t)bc * kp) + 16u * (unsigned)(p ^ ((rl >> 2) & 3));
        }
    }
    const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)bg_lds;
    const int krot = (int)((((blockIdx.x + gridDim.x * blockIdx.y) & 7u) * (unsigned)nk_) >> 3);   // per XCD: its workgroups share operands, and lines, in step
    auto dma = [&](int kstep) {   // K step `kstep` into its stage: rows [LR wave, +LR) of A and of B, 16 rows (1 KiB) a piece
        // the K steps in an order rotated per XCD (exact i32 sums: any order gives the same total): the eight XCDs do not all ask the
        // memory side for the same lines of A and B at the same moment; the workgroups of one XCD stay in step and share their L2 lines
        int kk = (kstep < nk_ ? kstep : nk_ - 1) + krot;
        kk = kk >= nk_ ? kk - nk_ : kk;
        const unsigned k0 = (unsigned)kk * BG_BK;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(kstep & 3) * BG_STAGE + (unsigned)(LR * wave) * BG_BK);
#pragma unroll
        for (int j = 0; j < LI; ++j) bg_dma16(a, aoff[j] + k0, dst + j * 16 * BG_BK);
#pragma unroll
        for (int j = 0; j < LI; ++j) bg_dma16(b, boff[j] + k0, dst + BG_BM * BG_BK + j * 16 * BG_BK);
    };
    const int nk = kp / BG_BK;   // even (kp is a multiple of 128)
    v4i sa0[LI], sb0[LI], sa1[LI], sb1[LI];   // two staging sets: K steps of even / odd number, requested two steps before they are stored
    auto fetch = [&](v4i (&sa)[LI], v4i (&sb)[LI], int kstep) {
        const unsigned k0 = (unsigned)(kstep < nk ? kstep : nk - 1) * BG_BK;   // unconditional: beyond the end the last step again, never stored
#pragma unroll
        for (int j = 0; j < LI; ++j) sa[j] = *reinterpret_cast<const v4i*>(reinterpret_cast<const char*>(a) + aoff[j] + k0);
#pragma unroll
        for (int j = 0; j < LI; ++j) sb[j] = *reinterpret_cast<const v4i*>(reinterpret_cast<const char*>(b) + boff[j] + k0);
    };
    auto put = [&](const v4i (&sa)[LI], const v4i (&sb)[LI], int stage) {
        char* const base = bg_lds + stage * BG_STAGE;
#pragma unroll
        for (int j = 0; j < LI; ++j) *reinterpret_cast<v4i*>(base + woff[j]) = sa[j];
#pragma unroll
        for (int j = 0; j < LI; ++j) *reinterpret_cast<v4i*>(base + BG_BM * BG_BK + woff[j]) = sb[j];
    };
    // ---- reader: lane (row l31 of a tile, half hv) takes chunk 2 s + hv of its row for k-step s: at position (2 s + hv) ^ swizzle
    const int swz = (l31 >> 2) & 3;
    const int roff0 = l31 * BG_BK + ((hv ^ swz) * 16), roff1 = l31 * BG_BK + (((2 + hv) ^ swz) * 16);
    v4i fa0[4], fb0[TJ], fa1[4], fb1[TJ];   // fragments of k-step 0 / 1 of a stage
    auto frags = [&](v4i (&fa)[4], v4i (&fb)[TJ], int stage, int roff) {
        const char* const abase = bg_lds + stage * BG_STAGE + (128 * wr) * BG_BK + roff;
        const char* const bbase = bg_lds + stage * BG_STAGE + BG_BM * BG_BK + (32 * TJ * wc) * BG_BK + roff;
#pragma unroll
        for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const v4i*>(abase + i * 32 * BG_BK);
#pragma unroll
        for (int j = 0; j < TJ; ++j) fb[j] = *reinterpret_cast<const v4i*>(bbase + j * 32 * BG_BK);
    };
    v16i acc[TJ][4];   // [column tile][row tile]
#pragma unroll
    for (int j = 0; j < TJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][i][r] = 0;
    auto products = [&](const v4i (&fa)[4], const v4i (&fb)[TJ]) {
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[j][i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb[j], fa[i], acc[j][i], 0, 0, 0);
    };
    // Schedule (one wave a SIMD: whatever the wave waits for, the matrix core waits for too, so nothing may be waited for that was not
    // asked for long ago).  FOUR stages of 64 bytes of K.  At step kt: stage kt is multiplied; stage kt + 1 is complete and visible
    // since the barrier that ended step kt - 1, so its first fragments are requested BEFORE this step's barrier; the operands of step
    // kt + 2 (requested from memory during step kt - 2) are stored into stage kt + 2, whose last readers passed the barrier of step
    // kt - 2; the operands of step kt + 4 are requested.  One raw s_barrier a step, in front of it only lgkmcnt(0) (the stores): the
    // global loads stay in flight across it.
    auto step = [&](int kt, v4i (&sa)[LI], v4i (&sb)[LI]) {
        const int st = kt & 3;
        frags(fa1, fb1, st, roff1);
        put(sa, sb, (kt + 2) & 3);
        products(fa0, fb0);
        // one LDS operation between two MFMAs, never a burst: a wave issues in order, and a burst of 1 KiB reads (times the other waves')
        // waits for the LDS queue with the matrix core idle behind it.  (Reads first, then stores: the compiler keeps LDS operations that
        // may alias in program order.)
        if constexpr (WC == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch(sa, sb, kt + 4);
        frags(fa0, fb0, (kt + 1) & 3, roff0);
        products(fa1, fb1);
        if constexpr (WC == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };
    if constexpr (DMA) {
        // At step kt: stage kt is multiplied, the first fragments of stage kt + 1 are requested (complete since the barrier that ended
        // step kt - 1), K step kt + 3 is requested from memory into the stage last read in step kt - 1; before the barrier that ends
        // step kt this wave's pieces of K step kt + 2 have landed (vmcnt: only the pieces of step kt + 3 may still be in flight).
        // TWO K steps between barriers (four stages = two pairs of stages used alternately).  At pair b (steps 2 b, 2 b + 1): their stages
        // have been complete since barrier b - 1 (every piece is waited for -- vmcnt(0) -- in front of a barrier); K steps 2 b + 2 and
        // 2 b + 3 are requested at the start of the pair into the two stages read in pair b - 1 and waited for at its end: two whole
        // steps between a request and its wait, half the barriers of the one-step form (which could ask for the next stage's first
        // fragments before its barrier; here they are asked for right behind it, under the next pair's eight requests).
        auto pair_dma = [&](int kt) {
            dma(kt + 2);
            dma(kt + 3);
            __builtin_amdgcn_s_setprio(1);
            frags(fa1, fb1, kt & 3, roff1);
            products(fa0, fb0);
#pragma unroll
            for (int q = 0; q < 4 + TJ; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            frags(fa0, fb0, (kt + 1) & 3, roff0);
            products(fa1, fb1);
#pragma unroll
            for (int q = 0; q < 4 + TJ; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            frags(fa1, fb1, (kt + 1) & 3, roff1);
            products(fa0, fb0);
#pragma unroll
            for (int q = 0; q < 4 + TJ; ++q) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            products(fa1, fb1);
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            frags(fa0, fb0, (kt + 2) & 3, roff0);   // the next pair's first fragments: its stages are complete now
        };
        dma(0);
        dma(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        frags(fa0, fb0, 0, roff0);
        for (int kt = 0; kt < nk; kt += 2) pair_dma(kt);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
    fetch(sa0, sb0, 0);
    fetch(sa1, sb1, 1);
    put(sa0, sb0, 0);
    fetch(sa0, sb0, 2);
    put(sa1, sb1, 1);
    fetch(sa1, sb1, 3);
    __syncthreads();   // (drains the two requests above as well: once)
    frags(fa0, fb0, 0, roff0);
    for (int kt = 0; kt < nk; kt += 2) {
        step(kt, sa0, sb0);
        step(kt + 1, sa1, sb1);
    }
    }
    // ---- epilogue: lane = result row 128 wr + 32 i + l31, columns 32 TJ wc + 32 j + 8 g + 4 hv + [0, 4)
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
        for (int j = 0; j < TJ; ++j) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = 32 * TJ * wc + 32 * j + 8 * g + 4 * hv;
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
