// gemm_core.h -- f32-input MFMA GEMM core for gfx950 (v_mfma_f32_32x32x2_f32: exact f32, k-ordered FMA chain).
//
// One kernel template serves every op that lele routes through faer's f32 matmul on x86
// (/root/reference/src/kernels/gemm.rs:196,315,527; conv1d.rs:1086,1259; conv2d.rs:358,686; rnn.rs:134-149):
// the operands are supplied by LOADER functors (row-major, column-major, implicit im2col ...) and the result is
// consumed by an EPILOGUE functor (plain store, alpha/beta*C, bias + ReLU/SiLU into NCHW ...).
//
// Tiling: block tile BM x BN, K step BK (16 or 32), WM x WN waves, each wave owns (BM/WM) x (BN/WN) as 32x32 MFMA tiles.
// Both operand tiles live in LDS as [row][BK k + 4 pad] (pitch 20 / 36 floats, so the two ds_read_b128 a lane
// issues per 32-row tile are 16-B aligned and a 16-lane group touches 16 distinct 16-B slots: 5*row mod 16).
// Lane l feeds MFMA step s (0..7) of sub-step u with k = 16*u + 8*(l>>5) + s of the K tile for BOTH operands, i.e. the
// K index is permuted identically on A and B -- a reordering of the exact f32 sum, nothing else.
// Global -> register -> LDS with the next tile's loads in flight during the MFMAs (double-buffered LDS,
// one __syncthreads per K tile).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
#include <stdlib.h>

namespace gemm {

// K depth of one LDS tile is a template parameter (16 or 32 floats); rows are padded by 4 floats (pitch 20 / 36), both
// of which keep 16-B alignment and give the 16 lanes of a ds_read_b128 group 16 distinct 16-B bank slots.

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------- loaders
// Loaders never put a global load behind a per-lane branch (that serialises one memory round trip per element):
// coordinates are clamped into range, the load is unconditional, and out-of-range elements are zeroed by a select.
//
// Loader protocol: `row(b, r)` does everything that does not depend on k ONCE per staging slot (clamping, the row's
// address, for convolutions the output position -> input window arithmetic); `get4(Row, k)` is what the K loop pays
// per step.  In the tiled kernel k is wave-uniform for row-fast loaders (64 consecutive threads share the k chunk), so
// whatever get4 derives from k alone lands on the scalar unit.  (Measured before the split: 10-12 vector instructions
// of index arithmetic per MFMA in the K loop; an MFMA 32x32x2 occupies the matrix pipe for 64 cycles = 16 of them.)
// element(row, k) = p[batch*bs + row*ld + k]   (k contiguous in memory)
struct LoadRowK {
    const float* p;
    int64_t bs;  // batch stride (0 = broadcast)
    int64_t ld;
    int rows, K;
    int vec;  // 1 when float4 loads are legal (base and ld 16-B aligned)
    // two-level batch (lele_hip_matmul_view: heads inside a packed tensor): b -> (b / binner) * bs + (b % binner) * bs2
    int binner = 1;
    int64_t bs2 = 0;
    static constexpr bool kRowFast = false;
    struct Row {
        const float* q;
        bool rin;
    };
    __device__ __forceinline__ int64_t boff(int b) const {
        return binner > 1 ? (int64_t)(b / binner) * bs + (int64_t)(b % binner) * bs2 : (int64_t)b * bs;
    }
    __device__ __forceinline__ Row row(int b, int r) const {
        LELE_DEV_ASSERT(b >= 0 && r >= 0 && rows >= 1 && K >= 1);
        const bool rin = r < rows;
        return Row{p + boff(b) + (int64_t)(rin ? r : rows - 1) * ld, rin};
    }
    __device__ __forceinline__ float4 get4(const Row& r, int k) const {
        float4 v;
        LELE_DEV_ASSERT(k >= 0);
        if (vec && k + 3 < K) {  // whole chunk in range (always, except in the last K tile)
            LELE_DEV_ASSERT((((uintptr_t)(r.q + k)) & 15) == 0);
            v = *reinterpret_cast<const float4*>(r.q + k);
        } else {
            const int last = K - 1;
            const float e0 = r.q[k + 0 < K ? k + 0 : last], e1 = r.q[k + 1 < K ? k + 1 : last];
            const float e2 = r.q[k + 2 < K ? k + 2 : last], e3 = r.q[k + 3 < K ? k + 3 : last];
            v = make_float4(k + 0 < K ? e0 : 0.f, k + 1 < K ? e1 : 0.f, k + 2 < K ? e2 : 0.f, k + 3 < K ? e3 : 0.f);
        }
        return r.rin ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ float4 get4(int b, int r, int k) const { return get4(row(b, r), k); }
};
// element(row, k) = p[batch*bs + k*ld + row]   (row contiguous in memory: B of a plain matmul, A of transA)
struct LoadKRow {
    const float* p;
    int64_t bs;
    int64_t ld;
    int rows, K;
    int binner = 1;  // two-level batch, as LoadRowK
    int64_t bs2 = 0;
    static constexpr bool kRowFast = true;
    struct Row {
        const float* base;  // p + batch offset: uniform over the workgroup
        unsigned off;       // clamped row
        bool rin;
    };
    __device__ __forceinline__ int64_t boff(int b) const {
        return binner > 1 ? (int64_t)(b / binner) * bs + (int64_t)(b % binner) * bs2 : (int64_t)b * bs;
    }
    __device__ __forceinline__ Row row(int b, int r) const {
        LELE_DEV_ASSERT(b >= 0 && r >= 0 && rows >= 1 && K >= 1);
        const bool rin = r < rows;
        return Row{p + boff(b), (unsigned)(rin ? r : rows - 1), rin};
    }
    __device__ __forceinline__ float4 get4(const Row& r, int k) const {
        const int last = K - 1;
        LELE_DEV_ASSERT(k >= 0 && r.off < (unsigned)rows);
        // (base + k*ld) is scalar arithmetic when k is uniform; the lane contributes only its 32-bit row offset
        const float e0 = (r.base + (int64_t)(k + 0 < K ? k + 0 : last) * ld)[r.off];
        const float e1 = (r.base + (int64_t)(k + 1 < K ? k + 1 : last) * ld)[r.off];
        const float e2 = (r.base + (int64_t)(k + 2 < K ? k + 2 : last) * ld)[r.off];
        const float e3 = (r.base + (int64_t)(k + 3 < K ? k + 3 : last) * ld)[r.off];
        return make_float4(r.rin && k + 0 < K ? e0 : 0.f, r.rin && k + 1 < K ? e1 : 0.f, r.rin && k + 2 < K ? e2 : 0.f,
                           r.rin && k + 3 < K ? e3 : 0.f);
    }
    __device__ __forceinline__ float4 get4(int b, int r, int k) const { return get4(row(b, r), k); }
};

// ---------------------------------------------------------------------------------------------- epilogues
enum CMode { C_NONE = 0, C_FULL = 1, C_ROWVEC = 2 /* [N] */, C_COLVEC = 3 /* [M] */, C_SCALAR = 4, C_MODULO = 5 };

// Epilogue protocol: `load(b, row, col)` fetches whatever the element needs from memory (C, bias ...) and is called
// with CLAMPED in-range coordinates, unconditionally, for all 16 accumulators of an MFMA tile before any of them is
// stored -- so the loads are issued back to back.  (A load behind a per-lane bounds branch compiles to a serialised
// branch + s_waitcnt per element: 64 dependent round trips per thread.)  `store(b, row, col, acc, pre)` does the
// bounds check and the arithmetic.
// out[b][row][col] = alpha*acc + beta*C[...]  (matmul, matmul_fused_add, gemm)
struct EpiAffine {
    float* out;
    int64_t bs;  // batch stride of out
    int M, N;
    float alpha, beta;
    const float* c;
    int cmode;
    int64_t clen;
    // strided output (lele_hip_matmul_view: the product stored straight into a transposed layout): row pitch ldo (0 = N),
    // two-level batch as the loaders.  The C operand is not combined with it.
    int64_t ldo = 0;
    int binner = 1;
    int64_t bs2 = 0;
    __device__ __forceinline__ int64_t ooff(int b, int row) const {
        const int64_t bo = binner > 1 ? (int64_t)(b / binner) * bs + (int64_t)(b % binner) * bs2 : (int64_t)b * bs;
        return bo + (int64_t)row * (ldo ? ldo : (int64_t)N);
    }
    __device__ __forceinline__ float load(int b, int row, int col) const {
        if (cmode == C_NONE) return 0.0f;  // uniform branch (kernel argument)
        int64_t idx;
        switch (cmode) {
            case C_FULL: idx = (int64_t)row * N + col; break;
            case C_ROWVEC: idx = col; break;
            case C_COLVEC: idx = row; break;
            case C_SCALAR: idx = 0; break;
            default: idx = ((int64_t)b * bs + (int64_t)row * N + col) % clen; break;
        }
        return c[idx];
    }
    // {min, max} of the tile a workgroup stores, one pair per workgroup (small-problem kernel only): a dynamic quantisation
    // that reads the whole result next reduces these pairs instead of scanning the tensor (common.h, LeleBuf::rowstat)
    float* blockstat = nullptr;
    __device__ __forceinline__ float value(float acc, float pre) const {
        return __builtin_fmaf(alpha, acc, cmode == C_NONE ? 0.0f : pre * beta);
    }
    __device__ __forceinline__ void store(int b, int row, int col, float acc, float pre) const {
        if (row >= M || col >= N) return;
        LELE_DEV_ASSERT(b >= 0 && row >= 0 && col >= 0);
        out[ooff(b, row) + col] = value(acc, pre);
    }
    // the 16-byte store protocol (has_vec_store): rows of the result may be written four columns at a time
    __device__ __forceinline__ bool vec_ok() const {
        return (N & 3) == 0 && ((ldo ? ldo : (int64_t)N) & 3) == 0 && (bs & 3) == 0 && (bs2 & 3) == 0 && ((uintptr_t)out & 15) == 0;
    }
    __device__ __forceinline__ float finish(int b, int row, int col, float acc, float pre) const { return value(acc, pre); }
    __device__ __forceinline__ float* row_ptr(int b, int row) const { return out + ooff(b, row); }
};

// epilogues that can publish per-workgroup {min, max} carry a `blockstat` member
template <class E, class = void>
struct has_blockstat : std::false_type {};
template <class E>
struct has_blockstat<E, std::void_t<decltype(std::declval<E>().blockstat)>> : std::true_type {};

// epilogues that can hand out finished values and take them back four columns at a time (`finish`, `row_ptr`, `vec_ok`): the
// tiled kernel then turns every 32 x 32 accumulator tile through LDS and stores 16 bytes per lane instead of 4 (a lane owns a
// COLUMN of the tile; stored as they sit, a tile is 16 four-byte store instructions per lane and the tail of a workgroup is
// store-issue bound)
// epilogues that take the FOUR consecutive rows a lane holds for one column in one call (`store_quad`, `quad_ok`): a transposed
// convolution with kernel = stride = 2 turns them into two 8-byte stores of horizontally adjacent outputs (conv.hip, ConvTKsEpi)
template <class E, class = void>
struct has_quad_store : std::false_type {};
template <class E>
struct has_quad_store<E, std::void_t<decltype(std::declval<E>().quad_ok())>> : std::true_type {};
// epilogues with a residual added to the finished values (`res_row_ptr`: the residual's row beside an output row, or NULL)
template <class E, class = void>
struct has_res_row : std::false_type {};
template <class E>
struct has_res_row<E, std::void_t<decltype(std::declval<E>().res_row_ptr(0, 0))>> : std::true_type {};
template <class E, class = void>
struct has_vec_store : std::false_type {};
template <class E>
struct has_vec_store<E, std::void_t<decltype(std::declval<E>().vec_ok())>> : std::true_type {};

// min / max of `mn`, `mx` over a 256-thread workgroup -> thread 0 writes the pair (qminmax_kernel's comparisons)
__device__ __forceinline__ void block_minmax_store(float mn, float mx, float* pair) {
    for (int off = 32; off > 0; off >>= 1) {
        const float a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    __shared__ float s_mn[4], s_mx[4];
    if ((threadIdx.x & 63) == 0) {
        s_mn[threadIdx.x >> 6] = mn;
        s_mx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = s_mn[w] < mn ? s_mn[w] : mn;
            mx = s_mx[w] > mx ? s_mx[w] : mx;
        }
        pair[0] = mn;
        pair[1] = mx;
    }
}

// ---------------------------------------------------------------------------------------------- tile order
// The dispatcher is observed to place workgroup b on XCD b % 8 (MI355X_MICROARCH.md, "Workgroup dispatch"; a matter of
// speed only, never of correctness), and every XCD has its own 4 MiB L2.  In launch order, the tiles that share an
// A row-panel or a B column-panel would therefore be spread over all eight L2s and each of them would fetch the panel
// from HBM again (measured on the batched 171x171x128 attention product: 64 % L2 misses, 3x the compulsory traffic).
// tile_coords() re-labels the workgroups so that every XCD works through one CONTIGUOUS range of logical tiles, and
// walks that range in patches of up to 8 tile-rows so that the tiles in flight together form a compact 2-D block.
__device__ __forceinline__ void tile_coords(unsigned& tx, unsigned& ty, unsigned& tz) {
    const unsigned gx = gridDim.x, gy = gridDim.y, per_batch = gx * gy, total = per_batch * gridDim.z;
    const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
    const unsigned xcd = lin & 7u, base = total >> 3, rem = total & 7u;
    const unsigned logical = xcd * base + (xcd < rem ? xcd : rem) + (lin >> 3);
    tz = logical / per_batch;
    const unsigned l = logical - tz * per_batch;
    const unsigned width = 8u * gx, group = l / width, first = group * 8u;
    const unsigned rows = gy - first < 8u ? gy - first : 8u;
    const unsigned in_group = l - group * width;
    ty = first + in_group % rows;
    tx = in_group / rows;
}

// ---------------------------------------------------------------------------------------------- kernel
template <int BM, int BN, int WM, int WN, int BK, class AL, class BL, class EPI, int OCC = 1>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(OCC))) void gemm_f32_mfma_kernel(AL al, BL bl, EPI epi, int M, int N, int K) {
    constexpr int NT = WM * WN * 64;
    constexpr int PITCH = BK + 4;
    constexpr int KQ = BK / 4;                              // float4 chunks per tile row
    constexpr int TMT = BM / WM / 32, TNT = BN / WN / 32;  // 32x32 MFMA tiles per wave
    constexpr int ASLOTS = (BM * KQ + NT - 1) / NT, BSLOTS = (BN * KQ + NT - 1) / NT;  // float4 staging slots per thread
    extern __shared__ __attribute__((aligned(16))) float gemm_lds[];  // As[2][BM*PITCH] then Bs[2][BN*PITCH]
    float(*As)[BM * PITCH] = reinterpret_cast<float(*)[BM * PITCH]>(gemm_lds);
    float(*Bs)[BN * PITCH] = reinterpret_cast<float(*)[BN * PITCH]>(gemm_lds + 2 * BM * PITCH);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    unsigned tx, ty, tz;
    tile_coords(tx, ty, tz);
    const int m0 = ty * BM, n0 = tx * BN, batch = tz;
    const int hv = lane >> 5, l31 = lane & 31;

    float4 ra[ASLOTS], rb[BSLOTS];
    // per staging slot, once: the loader's k-invariant row state, the k offset of the slot's chunk (made wave-uniform
    // where the slot -> (row, chunk) map allows it) and the LDS address the chunk is parked at
    typename AL::Row arow[ASLOTS];
    typename BL::Row brow[BSLOTS];
    int akq[ASLOTS], bkq[BSLOTS], alds[ASLOTS], blds[BSLOTS];
#pragma unroll
    for (int i = 0; i < ASLOTS; ++i) {
        const int s = tid + i * NT;
        int row = AL::kRowFast ? s % BM : s / KQ, kq = AL::kRowFast ? s / BM : s % KQ;
        if (AL::kRowFast && BM % 64 == 0) kq = __builtin_amdgcn_readfirstlane(kq);
        if (s >= BM * KQ) row = 0, kq = 0;
        arow[i] = al.row(batch, m0 + row);
        akq[i] = 4 * kq;
        alds[i] = row * PITCH + 4 * kq;
    }
#pragma unroll
    for (int i = 0; i < BSLOTS; ++i) {
        const int s = tid + i * NT;
        int row = BL::kRowFast ? s % BN : s / KQ, kq = BL::kRowFast ? s / BN : s % KQ;
        if (BL::kRowFast && BN % 64 == 0) kq = __builtin_amdgcn_readfirstlane(kq);
        if (s >= BN * KQ) row = 0, kq = 0;
        brow[i] = bl.row(batch, n0 + row);
        bkq[i] = 4 * kq;
        blds[i] = row * PITCH + 4 * kq;
    }
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i)
            if (tid + i * NT < BM * KQ) ra[i] = al.get4(arow[i], k0 + akq[i]);
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i)
            if (tid + i * NT < BN * KQ) rb[i] = bl.get4(brow[i], k0 + bkq[i]);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i)
            if (tid + i * NT < BM * KQ) *reinterpret_cast<float4*>(&As[buf][alds[i]]) = ra[i];
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i)
            if (tid + i * NT < BN * KQ) *reinterpret_cast<float4*>(&Bs[buf][blds[i]]) = rb[i];
    };

    f32x16 acc[TMT][TNT];
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int nk = (K + BK - 1) / BK;
    auto compute = [&](int cur) {
#pragma unroll
        for (int sub = 0; sub < BK / 16; ++sub) {  // 16 k per sub-step: fragment registers are reused
            float a[TMT][8], b[TNT][8];
#pragma unroll
            for (int i = 0; i < TMT; ++i) {
                const float* src = &As[cur][(wm * TMT * 32 + i * 32 + l31) * PITCH + sub * 16 + 8 * hv];
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                a[i][0] = v0.x; a[i][1] = v0.y; a[i][2] = v0.z; a[i][3] = v0.w;
                a[i][4] = v1.x; a[i][5] = v1.y; a[i][6] = v1.z; a[i][7] = v1.w;
            }
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                const float* src = &Bs[cur][(wn * TNT * 32 + j * 32 + l31) * PITCH + sub * 16 + 8 * hv];
                const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
                b[j][0] = v0.x; b[j][1] = v0.y; b[j][2] = v0.z; b[j][3] = v0.w;
                b[j][4] = v1.x; b[j][5] = v1.y; b[j][6] = v1.z; b[j][7] = v1.w;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int i = 0; i < TMT; ++i)
#pragma unroll
                    for (int j = 0; j < TNT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }
    };
    // tile kt+1 is in flight (global -> registers) during the MFMAs of tile kt.  (A second register stage -- loads two K
    // steps ahead -- was measured: no gain on long K loops, 3-12 % slower on the short ones.)
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BK);
        compute(cur);
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }
    // C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    if constexpr (has_vec_store<EPI>::value) {
        if (epi.vec_ok()) {  // uniform.  The operand buffers are free after the K loop's last barrier: 32 x 36 floats per wave
            static_assert((size_t)2 * (BM + BN) * PITCH >= (size_t)WM * WN * 32 * 36, "the operand buffers hold a tile per wave");
            float* mine = gemm_lds + wave * (32 * 36);
#pragma unroll
            for (int i = 0; i < TMT; ++i)
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    const int col = n0 + wn * TNT * 32 + j * 32 + l31;
                    const int colc = col < N ? col : N - 1;
                    const int rbase = m0 + wm * TMT * 32 + i * 32 + 4 * hv;
                    float pre[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = rbase + (r & 3) + 8 * (r >> 2);
                        pre[r] = epi.load(batch, row < M ? row : M - 1, colc);
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl = (r & 3) + 8 * (r >> 2) + 4 * hv;
                        mine[rl * 36 + l31] = epi.finish(batch, rbase - 4 * hv + rl < M ? rbase - 4 * hv + rl : M - 1, colc, acc[i][j][r], pre[r]);
                    }
                    const int c4 = n0 + wn * TNT * 32 + j * 32 + 4 * (lane & 7);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {  // same-wave LDS order holds: no barrier
                        const int rl = it * 8 + (lane >> 3), row = m0 + wm * TMT * 32 + i * 32 + rl;
                        float4 v = *reinterpret_cast<const float4*>(mine + rl * 36 + 4 * (lane & 7));
                        if constexpr (has_res_row<EPI>::value) {
                            const float* rp = epi.res_row_ptr(batch, row < M ? row : M - 1);
                            if (rp && row < M && c4 < N) {
                                const float4 r4 = *reinterpret_cast<const float4*>(rp + c4);
                                v.x += r4.x, v.y += r4.y, v.z += r4.z, v.w += r4.w;
                            }
                        }
                        LELE_DEV_ASSERT(!(row < M && c4 < N) || (c4 + 3 < N && (((uintptr_t)(epi.row_ptr(batch, row) + c4)) & 15) == 0));
                        if (row < M && c4 < N) *reinterpret_cast<float4*>(epi.row_ptr(batch, row) + c4) = v;  // N % 4 == 0: whole or absent
                    }
                }
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j) {
            const int col = n0 + wn * TNT * 32 + j * 32 + l31;
            const int colc = col < N ? col : N - 1;
            const int rbase = m0 + wm * TMT * 32 + i * 32 + 4 * hv;
            float pre[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                pre[r] = epi.load(batch, row < M ? row : M - 1, colc);
            }
            if constexpr (has_quad_store<EPI>::value) {
                if (epi.quad_ok()) {  // uniform
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float a4[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                        epi.store_quad(batch, rbase + 8 * q, col, a4, pre + 4 * q);
                    }
                    continue;
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) epi.store(batch, rbase + (r & 3) + 8 * (r >> 2), col, acc[i][j][r], pre[r]);
        }
}

// ---------------------------------------------------------------------------------------------- small problems
// One 256-thread workgroup per 32x32 output tile; its four waves split K four ways.  Every wave feeds its MFMAs straight
// from global memory through the SAME loader functors (the operands are L2-resident at these sizes): no LDS staging, no
// barrier in the K loop, all loads of a K step issued back to back.  The four partial tiles meet once in LDS and are
// added in the fixed order (p0+p1)+(p2+p3): deterministic, a re-association of the exact-product sum (inside 1e-4).
// Lane (l31, hv) owns k = 16c + 8hv + s of chunk c -- the operand mapping of the tiled kernel.
template <class AL, class BL, class EPI>
__global__ __launch_bounds__(256) void gemm_f32_small_kernel(AL al, BL bl, EPI epi, int M, int N, int K) {
    __shared__ float red[4][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int hv = lane >> 5, l31 = lane & 31;
    unsigned tx, ty, tz;
    tile_coords(tx, ty, tz);
    const int m0 = ty * 32, n0 = tx * 32, batch = tz;
    const int row = m0 + l31, col = n0 + l31;
    const int nchunk = (K + 15) / 16, per = (nchunk + 3) / 4;
    const int c0 = wave * per, c1 = c0 + per < nchunk ? c0 + per : nchunk;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int c = c0; c < c1; c += 2) {  // two 16-k chunks per trip: 4 get4 per operand in flight
        float4 fa[2][2], fb[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            // a chunk past this wave's range reads k >= K, which every loader returns as zeros
            const int k0 = c + u < c1 ? (c + u) * 16 + 8 * hv : K;
            fa[u][0] = al.get4(batch, row, k0);
            fa[u][1] = al.get4(batch, row, k0 + 4);
            fb[u][0] = bl.get4(batch, col, k0);
            fb[u][1] = bl.get4(batch, col, k0 + 4);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float a8[8] = {fa[u][0].x, fa[u][0].y, fa[u][0].z, fa[u][0].w, fa[u][1].x, fa[u][1].y, fa[u][1].z, fa[u][1].w};
            const float b8[8] = {fb[u][0].x, fb[u][0].y, fb[u][0].z, fb[u][0].w, fb[u][1].x, fb[u][1].y, fb[u][1].z, fb[u][1].w};
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a8[s], b8[s], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // wave w finishes accumulator registers 4w..4w+3 of the tile: row = (r&3) + 8*(r>>2) + 4*hv, col = l31
    const int colc = col < N ? col : N - 1;
    float pre[4], tot[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = 4 * wave + q;
        const int orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * hv;
        pre[q] = epi.load(batch, orow < M ? orow : M - 1, colc);
        tot[q] = (red[0][r][lane] + red[1][r][lane]) + (red[2][r][lane] + red[3][r][lane]);
    }
    bool quad_done = false;
    if constexpr (has_quad_store<EPI>::value) {
        if (epi.quad_ok()) {  // uniform: rows m0 + 8 wave + 4 hv + [0, 4) of column col
            epi.store_quad(batch, m0 + 8 * wave + 4 * hv, col, tot, pre);
            quad_done = true;
        }
    }
    if (!quad_done) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = 4 * wave + q;
            epi.store(batch, m0 + (r & 3) + 8 * (r >> 2) + 4 * hv, col, tot[q], pre[q]);
        }
    }
    if constexpr (has_blockstat<EPI>::value) {
        if (epi.blockstat) {  // uniform
            float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = 4 * wave + q;
                const int orow = m0 + (r & 3) + 8 * (r >> 2) + 4 * hv;
                if (orow < M && col < N) {
                    const float v = epi.value(tot[q], pre[q]);
                    mn = v < mn ? v : mn;
                    mx = v > mx ? v : mx;
                }
            }
            block_minmax_store(mn, mx, epi.blockstat + 2 * (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)));
        }
    }
}

// ---------------------------------------------------------------------------------------------- thin problems
// A result with at most 4 columns (THIN_M: at most 4 rows) is a handful of matrix-vector products: streaming chunk-by-chunk
// models (Silero: convolutions with 1-3 output positions, an LSTM step's W.x) consist of little else, and a 32x32 MFMA tile
// would be 90 % padding there.  One wave per row of the long side; its 64 lanes split K in float4 chunks read through the
// SAME loader functors (the long-side operand must be k-contiguous: !kRowFast), FMA into at most 4 accumulators, and meet
// in a shuffle butterfly -- a fixed order, so the result is deterministic; it is a re-association of the exact-product sum
// like every other kernel here (inside 1e-4).  No LDS, no barrier, every load of a lane issued before its first use.
template <bool THIN_M, class AL, class BL, class EPI>
__global__ __launch_bounds__(256) void gemm_f32_thin_kernel(AL al, BL bl, EPI epi, int M, int N, int K) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int batch = blockIdx.y;
    const int i = blockIdx.x * 4 + wave;  // row (column when THIN_M) of the long side: uniform over the wave
    if (i >= (THIN_M ? N : M)) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    auto dot = [](float s, const float4& a, const float4& b) {
        s = __builtin_fmaf(a.x, b.x, s);
        s = __builtin_fmaf(a.y, b.y, s);
        s = __builtin_fmaf(a.z, b.z, s);
        return __builtin_fmaf(a.w, b.w, s);
    };
    if constexpr (THIN_M) {
        const typename BL::Row lr = bl.row(batch, i);
        typename AL::Row fr[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) fr[f] = al.row(batch, f);  // rows past M load as zeros (loader protocol)
        for (int k = 4 * lane; k < K; k += 256) {
            const float4 l = bl.get4(lr, k);
            float4 v[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) v[f] = al.get4(fr[f], k);
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = dot(acc[f], v[f], l);
        }
    } else {
        const typename AL::Row lr = al.row(batch, i);
        typename BL::Row fr[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) fr[f] = bl.row(batch, f);
        for (int k = 4 * lane; k < K; k += 256) {
            const float4 l = al.get4(lr, k);
            float4 v[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) v[f] = bl.get4(fr[f], k);
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[f] = dot(acc[f], l, v[f]);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int f = 0; f < 4; ++f) acc[f] += __shfl_xor(acc[f], off);
    const int few = THIN_M ? M : N;
    if (lane < few) {
        const float mine = lane == 0 ? acc[0] : lane == 1 ? acc[1] : lane == 2 ? acc[2] : acc[3];
        const int row = THIN_M ? lane : i, col = THIN_M ? i : lane;
        epi.store(batch, row, col, mine, epi.load(batch, row, col));
    }
}

template <int BM, int BN, int WM, int WN, int BK, int OCC = 1, class AL, class BL, class EPI>
inline void launch_tile(hipStream_t st, const AL& al, const BL& bl, const EPI& epi, int M, int N, int K, int batch) {
    constexpr size_t lds = (size_t)2 * (BM + BN) * (BK + 4) * sizeof(float);
    auto kern = gemm_f32_mfma_kernel<BM, BN, WM, WN, BK, AL, BL, EPI, OCC>;
    if (lds > 64 * 1024)  // above the default per-block limit: opt in once per (instantiation, device)
        (void)lele::ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds);
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
    hipLaunchKernelGGL(kern, grid, dim3(WM * WN * 64), lds, st, al, bl, epi, M, N, K);
}

// Launch with the tile that wastes the least work: padded-tile efficiency (useful / computed elements) x the tile's
// intrinsic efficiency (bigger tiles reuse operands better), discounted when there are fewer workgroups than CUs.
// true when launch() will take the small-problem kernel (the only one that publishes epi.blockstat): its grid size
inline int64_t small_kernel_blocks(int M, int N, int K, int batch, int num_cus) {
    if (M <= 0 || N <= 0 || batch <= 0 || K < 16) return 0;
    if ((int64_t)((M + 63) / 64) * ((N + 63) / 64) * batch >= 2 * (int64_t)num_cus) return 0;
    return (int64_t)((N + 31) / 32) * ((M + 31) / 32) * batch;
}

template <class AL, class BL, class EPI>
inline void launch(hipStream_t st, const AL& al, const BL& bl, const EPI& epi, int M, int N, int K, int batch,
                   int num_cus) {
    if (M <= 0 || N <= 0 || batch <= 0) return;
    static const int force = lab_env("LELE_HIP_GEMM_FORCE") ? atoi(lab_env("LELE_HIP_GEMM_FORCE")) : -1;  // A/B experiments
    if (force >= 0) {
        switch (force) {
            case 0: launch_tile<128, 128, 2, 4, 16, 4>(st, al, bl, epi, M, N, K, batch); return;
            case 1: launch_tile<64, 256, 2, 4, 16, 4>(st, al, bl, epi, M, N, K, batch); return;
            case 2: launch_tile<64, 64, 2, 2, 16>(st, al, bl, epi, M, N, K, batch); return;
            case 3: launch_tile<32, 128, 1, 4, 16>(st, al, bl, epi, M, N, K, batch); return;
            case 4: launch_tile<256, 128, 4, 2, 16, 4>(st, al, bl, epi, M, N, K, batch); return;
            case 5: {
                dim3 grid((N + 31) / 32, (M + 31) / 32, batch);
                hipLaunchKernelGGL((gemm_f32_small_kernel<AL, BL, EPI>), grid, dim3(256), 0, st, al, bl, epi, M, N, K);
                return;
            }
            default: break;
        }
    }
    // at most 4 columns (or rows) of output: matrix-vector products, one wave per row of the long side (see above).  A caller
    // that wants the small-problem kernel's per-workgroup statistics keeps that kernel.
    bool wants_stats = false;
    if constexpr (has_blockstat<EPI>::value) wants_stats = epi.blockstat != nullptr;
    if (!wants_stats && K >= 8 && force < 0) {
        if (N <= 4 && M >= 8 && !AL::kRowFast) {
            hipLaunchKernelGGL((gemm_f32_thin_kernel<false, AL, BL, EPI>), dim3((M + 3) / 4, batch), dim3(256), 0, st, al, bl, epi, M, N, K);
            return;
        }
        if (M <= 4 && N >= 8 && !BL::kRowFast) {
            hipLaunchKernelGGL((gemm_f32_thin_kernel<true, AL, BL, EPI>), dim3((N + 3) / 4, batch), dim3(256), 0, st, al, bl, epi, M, N, K);
            return;
        }
    }
    // too few 64x64 tiles to fill the chip: 32x32 tiles with a 4-way split of K (latency-optimised, see above)
    if ((int64_t)((M + 63) / 64) * ((N + 63) / 64) * batch < 2 * (int64_t)num_cus && K >= 16) {
        dim3 grid((N + 31) / 32, (M + 31) / 32, batch);
        hipLaunchKernelGGL((gemm_f32_small_kernel<AL, BL, EPI>), grid, dim3(256), 0, st, al, bl, epi, M, N, K);
        return;
    }
    struct Cand {
        int bm, bn;
        double base;
    };
    // intrinsic efficiencies from measurement at 4096^3: 256x128 (8 waves) 114 TFLOP/s, 128x128 102, smaller tiles less
    static const Cand cands[5] = {{128, 128, 1.0}, {64, 256, 0.92}, {64, 64, 0.70}, {32, 128, 0.55}, {256, 128, 1.10}};
    int best = 0;
    double best_score = -1.0;
    for (int i = 0; i < 5; ++i) {
        const double tm = (M + cands[i].bm - 1) / cands[i].bm, tn = (N + cands[i].bn - 1) / cands[i].bn;
        const double eff = ((double)M * N) / (tm * cands[i].bm * tn * cands[i].bn);
        const double blocks = tm * tn * batch;
        const double fill = blocks >= 2.0 * num_cus ? 1.0 : blocks / (2.0 * num_cus);
        const double score = eff * cands[i].base * fill;
        if (score > best_score) {
            best_score = score;
            best = i;
        }
    }
    switch (best) {
        case 0: launch_tile<128, 128, 2, 4, 16, 4>(st, al, bl, epi, M, N, K, batch); break;  // 8 waves of 64x32
        case 1: launch_tile<64, 256, 2, 4, 16, 4>(st, al, bl, epi, M, N, K, batch); break;  // 8 waves of 32x64
        case 4: launch_tile<256, 128, 4, 2, 16, 4>(st, al, bl, epi, M, N, K, batch); break;  // 8 waves, 61 KB LDS, 2 per CU
        case 2: launch_tile<64, 64, 2, 2, 16>(st, al, bl, epi, M, N, K, batch); break;
        default: launch_tile<32, 128, 1, 4, 16>(st, al, bl, epi, M, N, K, batch); break;
    }
}

}  // namespace gemm
