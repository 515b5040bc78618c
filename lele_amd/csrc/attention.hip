// attention.hip -- softmax(Q K^T * scale) V for one packed projection, in ONE launch (emitted by lele_amd.compiler; never
// required by lele-generated code, which issues the node sequence this kernel replaces).
//
// Replaces, node for node, the sequence the compiler otherwise emits for an attention block of a SenseVoice-shaped encoder:
//     matmul_view(Q view, K^T view)                      /root/reference/src/kernels/gemm.rs:112-222   (matmul)
//     softmax_scaled(. , scale) = mul + softmax          src/kernels/math.rs:611, src/kernels/norm.rs:8 -> avx/norm.rs:139-229
//     matmul_view(P, V view) stored as [B, T, H, Dh]     gemm.rs:112-222 + manipulation.rs:644 (transpose) + shape.rs:2 (reshape)
// The [B, H, T, T] score and probability tensors never exist in HBM (2 x 15 MB per layer for BASELINE configs[3], 2 x 4 MB for
// configs[2]) and three launches become one.
//
// Arithmetic is the replaced sequence's, operation for operation:
//   * both products run on v_mfma_f32_32x32x2_f32 (exact f32 products, one FMA per term) with the SAME k order as the tiled GEMM
//     of gemm_core.h -- lane (l31, hv) owns k = 16c + 8hv + s of chunk c -- so S and O carry the bits the batched
//     `matmul_view` produces (the single-utterance `matmul_view` takes the K-split kernel, another order inside 1e-4);
//   * softmax is softmax_reg_kernel's row routine (norm_core.h): product with the scale rounded to f32 first, max, the
//     polynomial exp of the 8-wide body / libm tail, the sum in the 4 x 8 accumulator order, one division.
//
// Work split: one workgroup = 32 query rows of one (batch, head).  Phase 1: its four waves take the 32-key tiles round-robin;
// Q fragments live in registers (64 VGPRs), K fragments come straight from L2 in two register sets requested one set ahead
// (K and V of a head are 2 x T x 512 B: L2-resident and shared by the T / 32 workgroups of the head), S tiles go to LDS.
// Phase 2: eight 32-lane groups run the row softmax on four rows each, P overwrites S in LDS.  Phase 3: wave w owns output
// dims [32w, 32w + 32): A fragments = P from LDS, B fragments = V rows from L2, again one register set ahead.
// Each workgroup also leaves the {min, max} of what it stored next to the result, one pair per (utterance, head, row block):
// the output projection's dynamic quantisation needs no range pass (LeleBuf::rowstat kind 2, see quant.hip).
#include "common.h"
#include <type_traits>
#include "lane_ops.h"
#include "norm_core.h"

#include <stdlib.h>

using namespace lele;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct AttnArgs {
    const float* q;   // Q(b, i, d)   = q[bo * q_so + bi * q_si + i * q_sr + d]
    const float* k;   // K(b, j, d)   = k[bo * k_so + bi * k_si + j * k_sr + d]
    const float* v;   // V(b, j, d)   = v[bo * v_so + bi * v_si + j * v_sr + d]
    float* o;         // O(b, i, d)   = o[bo * o_so + bi * o_si + i * o_sr + d]
    int64_t q_so, q_si, q_sr, k_so, k_si, k_sr, v_so, v_si, v_sr, o_so, o_si, o_sr;
    int tq, tk, tpad;  // query rows, keys, keys rounded up to 64
    int batch_inner;   // heads (the inner batch dimension of the views)
    int nqb;           // query blocks per (batch, head)
    const float* scale;
    float* stat;       // [batch_outer][batch_inner * nqb][2] or NULL
#ifdef LELE_HIP_LAB
    int ablate;        // lab switch LELE_HIP_ATTN_ABLATE (timing experiments, results wrong): skip parts of the batch kernel
    long long* dbg;    // lab switch LELE_HIP_ATTN_STAMPS: cycle-counter stamps (tools/attention_stamps.py), or NULL
#endif
};

// ---- split-bf16 products (the default; LELE_HIP_ATTENTION_EXACT=1 keeps the f32 MFMA of the node sequence) ---------------------
// The f32-input MFMA runs at the f32 VECTOR rate -- 64 cycles per 32x32x2.  A bf16 MFMA does eight times the k extent in
// half the cycles, so an f32 value is cut into three bf16 pieces of 8 mantissa bits each (hi + mid + lo == x EXACTLY: truncation,
// then exact remainders) and a product becomes six bf16 MFMAs (hh, hm, mh, hl, lh, mm; the three dropped terms are <= 2^-24
// of the product: f32-rounding class).  Six 32-cycle instructions per 16 k against eight 64-cycle ones: 2.7x, on the matrix
// cores proper, with the vector pipe free for the split arithmetic and the softmax.  The result is the exact product's to ~1e-7
// relative -- inside the 1e-4 bar of the f32 GEMM family like every other summation order -- but not its bits.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Split3 {
    u32x4 h, m, l;  // eight values each, as bf16 pairs: element e in the low (even e) / high (odd e) half of word e / 2
};
__device__ __forceinline__ unsigned top16_pair(float even, float odd) {  // the upper halves of two f32 words side by side: one v_perm_b32
    return __builtin_amdgcn_perm(__float_as_uint(odd), __float_as_uint(even), 0x07060302u);
}
// Pieces by TRUNCATION here (mask, exact remainder), not rounded as the convolutions' are (common.h split3_bf16_pair): truncated pieces
// make every product ~1e-7 short in magnitude -- a bias that a 118-layer f32 convolution chain adds up coherently, and
// that means nothing where the result is re-quantised to 8 bits by the next linear, as every attention output of these models is.
// Measured: with rounded pieces (v_cvt_pk_bf16_f32 issues slower than v_and / v_perm) attention_flash_kernel 25.3 -> 27.5 us per
// configs[3] layer-shard, because here the MULTIPLYING waves split P themselves; the convolutions' loader waves split off the critical path.
__device__ __forceinline__ Split3 split3(const float (&x)[8]) {
    float r[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        r[e] = x[e] - __uint_as_float(__float_as_uint(x[e]) & 0xffff0000u);  // exact: the low 16 mantissa bits
        q[e] = r[e] - __uint_as_float(__float_as_uint(r[e]) & 0xffff0000u);  // exact: the last 8
    }
    Split3 s;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        s.h[p] = top16_pair(x[2 * p], x[2 * p + 1]);
        s.m[p] = top16_pair(r[2 * p], r[2 * p + 1]);
        s.l[p] = top16_pair(q[2 * p], q[2 * p + 1]);
    }
    return s;
}
#define LELE_BF(v) __builtin_bit_cast(bf16x8, v)
// acc += A . B over 16 k (32x32 tile): smallest terms first
__device__ __forceinline__ void mm6_32(const Split3& a, const Split3& b, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_BF(a.m), LELE_BF(b.m), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_BF(a.h), LELE_BF(b.l), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_BF(a.l), LELE_BF(b.h), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_BF(a.h), LELE_BF(b.m), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_BF(a.m), LELE_BF(b.h), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_BF(a.h), LELE_BF(b.h), acc, 0, 0, 0);
}

constexpr int kDh = 128;   // head dimension (8 chunks of 16)
constexpr int kSPad = 4;   // LDS row padding (floats)

// NT = softmax registers per lane (tpad <= 32 * NT).  RT = 32-row tiles of queries per workgroup (1 or 2).  With RT = 2 the waves
// pair up: wave w owns row tile w & 1 throughout; in phase 1 it takes every second key tile of that row tile (three each for the six
// tiles of a 10 s utterance: balanced, where RT = 1 leaves two of four waves with half the work), in phase 3 two of the four
// 32-dim output tiles, fed by ONE set of P fragments.  K and V are read half as often.
template <int NT, int RT, bool EXACT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void attention_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_sp[];  // S, then P: [32 * RT][tpad + 4]
    __shared__ float s_mm[4][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hv = lane >> 5, l31 = lane & 31;
    const int pitch = a.tpad + kSPad;
    const int qb = blockIdx.x % a.nqb, bh = blockIdx.x / a.nqb;
    const int bi = bh % a.batch_inner, bo = bh / a.batch_inner;
    const int i0 = qb * 32 * RT;
    const int rt = RT == 2 ? (wave & 1) : 0;   // this wave's row tile
    // its first key tile and the stride between its key tiles.  With one row tile the six key tiles of a 10 s utterance leave two
    // waves (= two SIMDs) with twice the work of the others: the workgroups that share a CU (blockIdx 256 apart) rotate the assignment
    const int kw = RT == 2 ? (wave >> 1) : ((wave - (int)(blockIdx.x >> 8)) & 3), kstep = 4 / RT;
    const float* qp = a.q + bo * a.q_so + bi * a.q_si;
    const float* kp = a.k + bo * a.k_so + bi * a.k_si;
    const float* vp = a.v + bo * a.v_so + bi * a.v_si;
    float* op = a.o + bo * a.o_so + bi * a.o_si;
    const int ntile = a.tpad / 32;  // key tiles (even: tpad is a multiple of 64)
    int nstamp = 0;
    auto stamp = [&]() {
#ifdef LELE_HIP_LAB
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 8 + nstamp++] = (long long)clock64();
#endif
    };
    stamp();

    // ---- phase 1: S = Q K^T.  Lane (l31, hv) of a 16-d chunk c owns d = 16c + 8hv + [0, 8): two float4 per chunk
    // K fragments in two register sets of half a head dimension each (4 chunks = 32 registers), requested one set ahead
    float4 ka[8], kb[8];
    int tq_ = kw, hq = 0;  // (key tile, half) of the next set to request
    auto reqk = [&](float4 (&w)[8]) {
        const int tc = tq_ < ntile ? tq_ : ntile - 1;                 // clamped: a harmless repeated load instead of a branch
        const int key = tc * 32 + l31;
        const float* src = kp + (int64_t)(key < a.tk ? key : a.tk - 1) * a.k_sr + 64 * hq + 8 * hv;  // padded keys re-read the last key
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            w[2 * c] = *reinterpret_cast<const float4*>(src + 16 * c);
            w[2 * c + 1] = *reinterpret_cast<const float4*>(src + 16 * c + 4);
        }
        hq ^= 1;
        if (hq == 0) tq_ += kstep;
    };
    reqk(ka);
    float4 qf[16];  // this lane's Q row: chunk c -> qf[2c], qf[2c + 1]
    {
        const int row = i0 + 32 * rt + l31;
        const float* src = qp + (int64_t)(row < a.tq ? row : a.tq - 1) * a.q_sr + 8 * hv;  // padded rows re-read the last row; never stored
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            qf[2 * c] = *reinterpret_cast<const float4*>(src + 16 * c);
            qf[2 * c + 1] = *reinterpret_cast<const float4*>(src + 16 * c + 4);
        }
    }
    // split-bf16 route: the query row's three pieces are made once (96 registers), the keys' per chunk as they arrive
    Split3 qs[EXACT ? 1 : 8];
    if constexpr (!EXACT) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float qa[8] = {qf[2 * c].x, qf[2 * c].y, qf[2 * c].z, qf[2 * c].w, qf[2 * c + 1].x, qf[2 * c + 1].y, qf[2 * c + 1].z, qf[2 * c + 1].w};
            qs[c] = split3(qa);
        }
    }
    auto mmk = [&](const float4 (&w)[8], int half, f32x16& acc) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float kk[8] = {w[2 * c].x, w[2 * c].y, w[2 * c].z, w[2 * c].w, w[2 * c + 1].x, w[2 * c + 1].y, w[2 * c + 1].z, w[2 * c + 1].w};
            if constexpr (EXACT) {
                const float4 q0 = qf[2 * (4 * half + c)], q1 = qf[2 * (4 * half + c) + 1];
                const float qa[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
#pragma unroll
                for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], kk[s], acc, 0, 0, 0);
            } else {
                mm6_32(qs[4 * half + c], split3(kk), acc);
            }
        }
    };
    for (int t = kw; t < ntile; t += kstep) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        reqk(kb);
        __builtin_amdgcn_sched_barrier(0);
        mmk(ka, 0, acc);
        __builtin_amdgcn_sched_barrier(0);
        reqk(ka);
        __builtin_amdgcn_sched_barrier(0);
        mmk(kb, 1, acc);
        __builtin_amdgcn_sched_barrier(0);
        // C layout of the 32x32 MFMA: column (key) = l31, row = (r & 3) + 8 (r >> 2) + 4 hv
        float* dst = s_sp + 32 * rt * pitch + t * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[((r & 3) + 8 * (r >> 2) + 4 * hv) * pitch] = acc[r];
    }
    stamp();  // [1] this wave's score tiles are in LDS
    __syncthreads();
    stamp();  // [2] every wave's

    // ---- phase 2: row softmax, eight 32-lane groups x 4 RT rows.  P = 0 beyond the last key (those columns then add exact zeros)
    {
        const int g = tid >> 5, l = tid & 31;
        const float sc = a.scale ? a.scale[0] : 1.0f;
        // four rows at a time: their reductions are independent chains of cross-lane operations (each a round trip through the LDS
        // crossbar), interleaved by the scheduler instead of queued one row after the other
#pragma unroll 1
        for (int r4 = 0; r4 < RT; ++r4) {
            float v[4][NT];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* row = s_sp + (g * 4 * RT + 4 * r4 + q) * pitch;
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const int j = 32 * c + l;
                    v[q][c] = row[j < a.tk ? j : a.tk - 1];
                    if (a.scale) v[q][c] = v[q][c] * sc;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (EXACT) softmax_row_reg<NT>(v[q], a.tk, l);
                else softmax_row_fast<NT>(v[q], a.tk, l);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float* row = s_sp + (g * 4 * RT + 4 * r4 + q) * pitch;
#pragma unroll
                for (int c = 0; c < NT; ++c) {
                    const int j = 32 * c + l;
                    if (j < a.tpad) row[j] = j < a.tk ? v[q][c] : 0.0f;
                }
            }
        }
    }
    stamp();  // [3] softmax of this group's rows
    __syncthreads();
    stamp();  // [4]

    // ---- phase 3: O = P V.  A wave owns ND = RT output tiles of 32 dims of its row tile; a set = 32 keys (two 16-key chunks):
    // lane (l31, hv) holds V[16c + 8hv + s][32 d + l31] (16 registers per output tile) and reads P[row l31][16c + 8hv .. + 8] from LDS
    constexpr int ND = RT;
    const int dw = RT == 2 ? (wave >> 1) : wave;  // first output tile; the second (RT = 2) is dw + 2
    float va[ND][16], vb[ND][16];
    int jq = 0;  // first key of the next set to request
    auto reqv = [&](float (&w)[ND][16]) {
        const int jc = jq < a.tpad ? jq : a.tpad - 32;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int key = jc + 16 * c + 8 * hv + s;
                const float* src = vp + (int64_t)(key < a.tk ? key : a.tk - 1) * a.v_sr + 32 * dw + l31;
#pragma unroll
                for (int d = 0; d < ND; ++d) w[d][8 * c + s] = src[64 * d];
            }
        jq += 32;
    };
    const float* prow = s_sp + (32 * rt + l31) * pitch + 8 * hv;
    f32x16 oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[d][r] = 0.0f;
    auto mmv = [&](const float (&w)[ND][16], int j0) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const float4 p0 = *reinterpret_cast<const float4*>(prow + j0 + 16 * c), p1 = *reinterpret_cast<const float4*>(prow + j0 + 16 * c + 4);
            const float pa[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
            if constexpr (EXACT) {
#pragma unroll
                for (int s = 0; s < 8; ++s)
#pragma unroll
                    for (int d = 0; d < ND; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], w[d][8 * c + s], oacc[d], 0, 0, 0);
            } else {
                const Split3 ps = split3(pa);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const float vv[8] = {w[d][8 * c], w[d][8 * c + 1], w[d][8 * c + 2], w[d][8 * c + 3], w[d][8 * c + 4], w[d][8 * c + 5], w[d][8 * c + 6], w[d][8 * c + 7]};
                    mm6_32(ps, split3(vv), oacc[d]);
                }
            }
        }
    };
    reqv(va);
    for (int j0 = 0; j0 < a.tpad; j0 += 64) {
        reqv(vb);
        __builtin_amdgcn_sched_barrier(0);
        mmv(va, j0);
        __builtin_amdgcn_sched_barrier(0);
        reqv(va);
        __builtin_amdgcn_sched_barrier(0);
        mmv(vb, j0 + 32);
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp();  // [5] P V products done
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * hv;
            if (row < a.tq) {
                const float val = oacc[d][r];
                op[(int64_t)row * a.o_sr + 32 * (dw + 2 * d) + l31] = val;
                mn = val < mn ? val : mn;
                mx = val > mx ? val : mx;
            }
        }
    stamp();  // [6] stores issued
    if (a.stat) {  // uniform: one pair per workgroup
        for (int off = 32; off > 0; off >>= 1) {
            const float p = __shfl_xor(mn, off), q = __shfl_xor(mx, off);
            mn = p < mn ? p : mn;
            mx = q > mx ? q : mx;
        }
        if (lane == 0) {
            s_mm[wave][0] = mn;
            s_mm[wave][1] = mx;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < 4; ++w) {
                mn = s_mm[w][0] < mn ? s_mm[w][0] : mn;
                mx = s_mm[w][1] > mx ? s_mm[w][1] : mx;
            }
            float* dst = a.stat + ((int64_t)bo * (a.batch_inner * a.nqb) + bi * a.nqb + qb) * 2;
            dst[0] = mn;
            dst[1] = mx;
        }
    }
}

// ---- small grids (a single utterance): 16 query rows per workgroup on the 16x16x4 MFMA ----------------------------------------
// One 30 s utterance has 4 heads x 16 blocks of 32 rows = 64 workgroups: three quarters of the chip idle, and each workgroup's
// chain of 64-cycle MFMAs is long.  Halving the block (128 workgroups) halves every phase.  The 16x16x4 instruction sums four k
// values per step, so a dot product is accumulated in another order than in the 32-row kernel and in the batched GEMM -- inside
// the 1e-4 bar of the f32 GEMM family (the sequence this replaces at this size, the K-split small-problem kernel, is likewise
// "another order"); softmax is the same routine, bit for bit on equal inputs.
// Lane l = (r = l & 15, g = l >> 4).  Scores: lane group g owns d in [32 g, 32 g + 32) for BOTH operands (so a step pairs equal
// d); Q row r in 8 float4 registers, a key tile = 16 keys, fragments requested one tile ahead.  P V: lane group g owns the keys
// [g tpad / 4, (g + 1) tpad / 4); a wave owns two 16-wide output tiles fed by one set of P fragments.
typedef float f32x4v __attribute__((ext_vector_type(4)));

// NW = waves per workgroup (4; with 8 a wave owns one 16-wide output tile and a 32-lane group one softmax row)
// acc += A . B over 32 k (16x16 tile: lane group g supplies k-slice g of the instruction's 32, eight values per lane)
__device__ __forceinline__ void mm6_16(const Split3& a, const Split3& b, f32x4v& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LELE_BF(a.m), LELE_BF(b.m), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LELE_BF(a.h), LELE_BF(b.l), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LELE_BF(a.l), LELE_BF(b.h), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LELE_BF(a.h), LELE_BF(b.m), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LELE_BF(a.m), LELE_BF(b.h), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(LELE_BF(a.h), LELE_BF(b.h), acc, 0, 0, 0);
}

template <int NT, int NW, bool EXACT>
__global__ __launch_bounds__(64 * NW) void attention16_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) float s_sp[];  // S, then P: [16][tpad + 4]
    __shared__ float s_mm[NW][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int pitch = a.tpad + kSPad;
    const int qb = blockIdx.x % a.nqb, bh = blockIdx.x / a.nqb;
    const int bi = bh % a.batch_inner, bo = bh / a.batch_inner;
    const int i0 = qb * 16;
    const float* qp = a.q + bo * a.q_so + bi * a.q_si;
    const float* kp = a.k + bo * a.k_so + bi * a.k_si;
    const float* vp = a.v + bo * a.v_so + bi * a.v_si;
    float* op = a.o + bo * a.o_so + bi * a.o_si;
    const int ntile = a.tpad / 16;
    int nstamp = 0;
    auto stamp = [&]() {
#ifdef LELE_HIP_LAB
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 8 + nstamp++] = (long long)clock64();
#endif
    };
    stamp();

    // ---- phase 1: S = Q K^T
    float4 ka[8], kb[8];
    int tnext = wave;
    auto reqk = [&](float4 (&w)[8]) {
        const int tc = tnext < ntile ? tnext : ntile - 1;  // clamped: a harmless repeated load instead of a branch
        const int key = tc * 16 + r16;
        const float* src = kp + (int64_t)(key < a.tk ? key : a.tk - 1) * a.k_sr + 32 * g;  // padded keys re-read the last key
#pragma unroll
        for (int c = 0; c < 8; ++c) w[c] = *reinterpret_cast<const float4*>(src + 4 * c);
        tnext += NW;
    };
    reqk(ka);
    float4 qf[8];
    {
        const int row = i0 + r16;
        const float* src = qp + (int64_t)(row < a.tq ? row : a.tq - 1) * a.q_sr + 32 * g;  // padded rows re-read the last row; never stored
#pragma unroll
        for (int c = 0; c < 8; ++c) qf[c] = *reinterpret_cast<const float4*>(src + 4 * c);
    }
    // split-bf16 route: instruction c of four pairs, in lane group g, the dims 32 g + 8 c + [0, 8) of both operands (any assignment
    // of k to (instruction, lane group, position) is a valid order as long as Q and K share it)
    Split3 qs[EXACT ? 1 : 4];
    if constexpr (!EXACT) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float qa[8] = {qf[2 * c].x, qf[2 * c].y, qf[2 * c].z, qf[2 * c].w, qf[2 * c + 1].x, qf[2 * c + 1].y, qf[2 * c + 1].z, qf[2 * c + 1].w};
            qs[c] = split3(qa);
        }
    }
    auto mmk = [&](const float4 (&w)[8], int t) {
        f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if constexpr (EXACT) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].x, w[c].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].y, w[c].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].z, w[c].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].w, w[c].w, acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float kk[8] = {w[2 * c].x, w[2 * c].y, w[2 * c].z, w[2 * c].w, w[2 * c + 1].x, w[2 * c + 1].y, w[2 * c + 1].z, w[2 * c + 1].w};
                mm6_16(qs[c], split3(kk), acc);
            }
        }
        // C layout of the 16x16 MFMA: column (key) = r16, row = 4 g + register
        float* dst = s_sp + (4 * g) * pitch + t * 16 + r16;
#pragma unroll
        for (int r = 0; r < 4; ++r) dst[r * pitch] = acc[r];
    };
    for (int t = wave; t < ntile; t += 2 * NW) {  // two tiles per trip: fragments of the next tile travel during the current one
        reqk(kb);
        __builtin_amdgcn_sched_barrier(0);
        mmk(ka, t);
        __builtin_amdgcn_sched_barrier(0);
        reqk(ka);
        __builtin_amdgcn_sched_barrier(0);
        if (t + NW < ntile) mmk(kb, t + NW);
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp();
    __syncthreads();
    stamp();

    // ---- phase 2: row softmax, 2 NW 32-lane groups x 16 / (2 NW) rows.  P = 0 beyond the last key (those columns then add exact zeros)
    {
        const int gi = tid >> 5, l = tid & 31;
        const float sc = a.scale ? a.scale[0] : 1.0f;
        constexpr int RPG = 16 / (2 * NW);  // rows per group
        float v[RPG][NT];
#pragma unroll
        for (int q = 0; q < RPG; ++q) {
            const float* row = s_sp + (gi * RPG + q) * pitch;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int j = 32 * c + l;
                v[q][c] = row[j < a.tk ? j : a.tk - 1];
                if (a.scale) v[q][c] = v[q][c] * sc;
            }
        }
#pragma unroll
        for (int q = 0; q < RPG; ++q) {
            if constexpr (EXACT) softmax_row_reg<NT>(v[q], a.tk, l);
            else softmax_row_fast<NT>(v[q], a.tk, l);
        }
#pragma unroll
        for (int q = 0; q < RPG; ++q) {
            float* row = s_sp + (gi * RPG + q) * pitch;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int j = 32 * c + l;
                if (j < a.tpad) row[j] = j < a.tk ? v[q][c] : 0.0f;
            }
        }
    }
    stamp();
    __syncthreads();
    stamp();

    // ---- phase 3: O = P V.  Wave w owns output dims [32 w, 32 w + 32) as two 16-wide tiles; lane group g sums over its quarter of
    // the keys: step c of a set pairs P[row r16][g kk + c] with V[g kk + c][32 w + 16 t + r16]
    const int kk = a.tpad / 4;  // keys per lane group: a multiple of 16
    const int kbase = g * kk;
    constexpr int ND = 8 / NW;       // 16-wide output tiles per wave
    const int d0 = 16 * ND * wave;    // first output dim of this wave
    float va[ND][16], vb[ND][16];
    int cq = 0;  // first step of the next set to request
    auto reqv = [&](float (&w)[ND][16]) {
        const int cc = cq < kk ? cq : kk - 16;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int key = kbase + cc + u;
            const float* src = vp + (int64_t)(key < a.tk ? key : a.tk - 1) * a.v_sr + d0 + r16;
#pragma unroll
            for (int d = 0; d < ND; ++d) w[d][u] = src[16 * d];
        }
        cq += 16;
    };
    const float* prow = s_sp + r16 * pitch + kbase;
    f32x4v oacc[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d) oacc[d] = f32x4v{0.0f, 0.0f, 0.0f, 0.0f};
    auto mmv = [&](const float (&w)[ND][16], int c0) {
        if constexpr (EXACT) {
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const float4 p = *reinterpret_cast<const float4*>(prow + c0 + 4 * u4);
                const float pa[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int d = 0; d < ND; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[e], w[d][4 * u4 + e], oacc[d], 0, 0, 0);
            }
        } else {  // the lane group's 16 keys of the set as two instructions of 8
#pragma unroll
            for (int u8 = 0; u8 < 2; ++u8) {
                const float4 p0 = *reinterpret_cast<const float4*>(prow + c0 + 8 * u8), p1 = *reinterpret_cast<const float4*>(prow + c0 + 8 * u8 + 4);
                const float pa[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
                const Split3 ps = split3(pa);
#pragma unroll
                for (int d = 0; d < ND; ++d) {
                    const float vv[8] = {w[d][8 * u8], w[d][8 * u8 + 1], w[d][8 * u8 + 2], w[d][8 * u8 + 3], w[d][8 * u8 + 4], w[d][8 * u8 + 5], w[d][8 * u8 + 6], w[d][8 * u8 + 7]};
                    mm6_16(ps, split3(vv), oacc[d]);
                }
            }
        }
    };
    reqv(va);
    for (int c0 = 0; c0 < kk; c0 += 32) {
        reqv(vb);
        __builtin_amdgcn_sched_barrier(0);
        mmv(va, c0);
        __builtin_amdgcn_sched_barrier(0);
        reqv(va);
        __builtin_amdgcn_sched_barrier(0);
        if (c0 + 16 < kk) mmv(vb, c0 + 16);
        __builtin_amdgcn_sched_barrier(0);
    }
    stamp();
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = i0 + 4 * g + r;
        if (row < a.tq) {
            float* dst = op + (int64_t)row * a.o_sr + d0 + r16;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                const float val = oacc[d][r];
                dst[16 * d] = val;
                mn = val < mn ? val : mn;
                mx = val > mx ? val : mx;
            }
        }
    }
    stamp();
    if (a.stat) {  // uniform: one pair per workgroup
        mn = wave_allreduce64(mn, [](float cur, float x) { return x < cur ? x : cur; });
        mx = wave_allreduce64(mx, [](float cur, float x) { return x > cur ? x : cur; });
        if (lane == 0) {
            s_mm[wave][0] = mn;
            s_mm[wave][1] = mx;
        }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NW; ++w) {
                mn = s_mm[w][0] < mn ? s_mm[w][0] : mn;
                mx = s_mm[w][1] > mx ? s_mm[w][1] : mx;
            }
            float* dst = a.stat + ((int64_t)bo * (a.batch_inner * a.nqb) + bi * a.nqb + qb) * 2;
            dst[0] = mn;
            dst[1] = mx;
        }
    }
}

// ---- large grids (a batch of utterances): one pass over the keys, nothing but registers between the two products ----------------
// The kernels above put S in LDS between three barrier-separated phases and let every lane fetch its key row with a 6 KB lane
// stride; with the products on split-bf16 their life is those loads and the operand splits (stamps: 20 k of 41 k cycles in the
// score phase for 2.3 k of MFMA).  This one is organised around what the matrix cores want:
//   * S^T = K Q^T and O^T = V^T P^T, so that the lane index is the QUERY in every accumulator (C layout: lane = column): a lane
//     owns query i for 16 keys of the tile (S^T), then supplies exactly those 16 probabilities as the B operand of the second
//     product (its k order is ours to choose: V's fragments are built in the same key order), and owns query i for 64 output
//     dimensions (O^T).  The row maximum / sum of a query live in ONE lane pair: no LDS round trip, no 32-lane butterflies, and
//     the running rescale of the online softmax is one multiplier per lane;
//   * a workgroup = four compute waves (four 32-row blocks of one head) + four PRODUCER waves, one of each per SIMD.  The
//     producers fetch the head's K and V tiles (32 keys) from global memory into registers one tile ahead, split every value
//     into its three bf16 pieces ONCE per workgroup, and store the pieces in LDS in MFMA fragment order (1 KiB = the 16 bytes
//     of each of the 64 lanes): a compute wave's operand is three conflict-free ds_read_b128, and its instruction stream is
//     MFMA + softmax only -- the splits issue on the same SIMD's VALU from the other wave while the matrix core runs;
//   * two LDS stages of {K pieces, V pieces} (48 KiB each), one s_barrier per key tile;
//   * softmax is the online form (running maximum m, running sum l, O rescaled by exp(m_old - m_new) per tile), on v_exp_f32.
// Same values as the node sequence to ~1e-6 relative (tests/test_attention.py holds it to the oracle at 2e-4 like the other
// kernels), not its bits: LELE_HIP_ATTENTION_EXACT=1 keeps the replica above.
constexpr int FA_PART = 8 * 3 * 1024, FA_STAGE = 2 * FA_PART, FA_LDS = 2 * FA_STAGE;  // 8 fragments x 3 pieces x 1 KiB, K then V
constexpr int FA_ROWS = 128;                                                          // query rows of a workgroup

#ifdef LELE_HIP_LAB
#define FA_STAMP(n_) \
    if (a.dbg && lane == 0) a.dbg[((size_t)blockIdx.x * 8 + wave) * 64 + (n_)] = (long long)clock64()
#else
#define FA_STAMP(n_)
#endif

__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void attention_flash_kernel(AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char fa_lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hv = lane >> 5, l31 = lane & 31;
    // workgroup b runs on XCD b % 8 (each with its own L2): the nqb workgroups of a head are 8 apart in b, so the second one's K / V
    // come out of the L2 the first one filled (heads in groups of 8; a ragged last group keeps the plain order)
    int qb = blockIdx.x % a.nqb, bh = blockIdx.x / a.nqb;
    {
        const int per = 8 * a.nqb, grp = blockIdx.x / per, heads = (int)(gridDim.x / a.nqb);
        if ((grp + 1) * 8 <= heads) {
            const int in = blockIdx.x - grp * per;
            bh = grp * 8 + (in & 7);
            qb = in >> 3;
        }
    }
    const int bi = bh % a.batch_inner, bo = bh / a.batch_inner;
    const int nkt = (a.tk + 31) / 32;
    // (a raw s_barrier: the compiler's own would wait vmcnt(0), i.e. for every load in flight.  What it must wait for is this wave's LDS
    // traffic -- a parked chunk's ds_writes are only ISSUED when the instruction after them runs, and the barrier hands the stage over)
    auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    if (wave >= 4) {
        // ------------------------------------------------------------ a producer: four of a tile's sixteen operand fragments
        const int pw = wave - 4;          // 0, 1: K chunks 4 pw .. 4 pw + 3;  2, 3: V^T fragments of key half c2 = pw - 2
        const bool is_k = pw < 2;
        const float* kp = a.k + bo * a.k_so + bi * a.k_si + 8 * hv + 64 * pw;  // K: lane = key l31, dims 16 c + 8 hv + [0, 8)
        const float* vp = a.v + bo * a.v_so + bi * a.v_si + l31;               // V^T: lane = dim 32 d + l31, keys 16 c2 + 4 hv + {0..3, 8..11}
        const int c2 = pw - 2;
        char* const part = fa_lds + (is_k ? pw * 4 : 8 + c2 * 4) * 3072 + lane * 16;
        // K producers and V producers run the SAME loop from two instantiations, chosen once: with `if (is_k)` inside fetch() every
        // request sat under a branch, and behind the merge the compiler's in-order count of loads in flight was gone -- a tile's split
        // waited vmcnt(0..3), i.e. for the tile requested a moment ago, in four of the six unrolled steps
        auto produce = [&](auto isk) {
            constexpr bool IS_K = decltype(isk)::value;
            float r0[4][8], r1[4][8], r2[4][8];  // three tiles of this wave's fragments: one being split, two in flight
            auto fetch = [&](float (&r)[4][8], int t) {
                t = t < nkt ? t : nkt - 1;  // past the end: the last tile again (never stored anywhere a compute wave reads)
                if constexpr (IS_K) {
                    int key = 32 * t + l31;
                    key = key < a.tk ? key : a.tk - 1;  // keys beyond the last re-read it: masked in the softmax
                    const float* src = kp + (int64_t)key * a.k_sr;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float4 k0 = *reinterpret_cast<const float4*>(src + 16 * c), k1 = *reinterpret_cast<const float4*>(src + 16 * c + 4);
                        r[c][0] = k0.x, r[c][1] = k0.y, r[c][2] = k0.z, r[c][3] = k0.w, r[c][4] = k1.x, r[c][5] = k1.y, r[c][6] = k1.z, r[c][7] = k1.w;
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        int key = 32 * t + 16 * c2 + 4 * hv + (e & 3) + 8 * (e >> 2);
                        key = key < a.tk ? key : a.tk - 1;
                        const float* src = vp + (int64_t)key * a.v_sr;
#pragma unroll
                        for (int d = 0; d < 4; ++d) r[d][e] = src[32 * d];
                    }
                }
            };
            auto put = [&](const float (&r)[4][8], int stage) {
#ifdef LELE_HIP_LAB
                if (a.ablate & 4) return;
#endif
                char* dst = part + stage * FA_STAGE;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const Split3 sp = split3(r[f]);
                    *reinterpret_cast<u32x4*>(dst + f * 3072) = sp.h;
                    *reinterpret_cast<u32x4*>(dst + f * 3072 + 1024) = sp.m;
                    *reinterpret_cast<u32x4*>(dst + f * 3072 + 2048) = sp.l;
                }
            };
            FA_STAMP(0);
            // tile 0 alone first: every workgroup of the grid starts at once, and what they ask for in their first microsecond is
            // served at HBM speed -- Q and the first tile are all the first product needs (stamps: with three tiles requested up
            // front the first operand reached LDS 14 k cycles into a 50 k cycle kernel)
            fetch(r0, 0);
            FA_STAMP(1);
            put(r0, 0);
            fetch(r1, 1);
            fetch(r2, 2);
            FA_STAMP(2);
            barrier();  // stage 0 holds tile 0
            FA_STAMP(3);
            // tile t (compute waves on stage t & 1): refill the registers tile t came from with tile t + 3, split tile t + 1 into
            // the other stage (its loads were issued two tiles ago).  Past the end the clamped tile lands where nobody reads.
#define LELE_FA_STEP(FREE_, NEXT_, STAGE_) \
    {                                      \
        fetch(FREE_, t + 3);               \
        FA_STAMP(4 + 3 * t);               \
        put(NEXT_, STAGE_);                \
        FA_STAMP(5 + 3 * t);               \
        barrier();                         \
        FA_STAMP(6 + 3 * t);               \
        if (++t >= nkt) break;             \
    }
            for (int t = 0;;) {
                LELE_FA_STEP(r0, r1, 1)
                LELE_FA_STEP(r1, r2, 0)
                LELE_FA_STEP(r2, r0, 1)
                LELE_FA_STEP(r0, r1, 0)
                LELE_FA_STEP(r1, r2, 1)
                LELE_FA_STEP(r2, r0, 0)
            }
#undef LELE_FA_STEP
        };
        if (is_k) produce(std::true_type{});
        else produce(std::false_type{});
        return;
    }
    // ---------------------------------------------------------------- a compute wave: 32 query rows of the head
    const int i0 = (qb * 4 + wave) * 32;
    const bool live = i0 < a.tq;  // a row block past the last query only keeps the barriers company
    const int row = i0 + l31;
    const bool rok = row < a.tq;
    // the scores in the exponent's own unit: Q carries scale * log2(e) into the product (one rounding of q instead of one of s)
    const float sc = (a.scale ? a.scale[0] : 1.0f) * 1.44269504088896341f;
    Split3 qs[8];
    FA_STAMP(0);
    if (live) {
        const float* src = a.q + bo * a.q_so + bi * a.q_si + (int64_t)(rok ? row : a.tq - 1) * a.q_sr + 8 * hv;  // padded rows re-read the last; never stored
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float4 q0 = *reinterpret_cast<const float4*>(src + 16 * c), q1 = *reinterpret_cast<const float4*>(src + 16 * c + 4);
            const float qa[8] = {q0.x * sc, q0.y * sc, q0.z * sc, q0.w * sc, q1.x * sc, q1.y * sc, q1.z * sc, q1.w * sc};
            qs[c] = split3(qa);
        }
    }
    f32x16 o[4];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.0f;
    // m_run: the exponent offset in use (log2 units).  It follows the running maximum only when that has grown by more than 8
    // since the last move: probabilities then reach 2^8 at most -- harmless in f32 and in the bf16 pieces -- and the rescale of
    // O and l, 66 multiplies a lane, happens on the first tile and then hardly ever instead of on every tile.  The final
    // division by l makes the offset's value immaterial.
    float m_run = -3.40282347e+38f, l_run = 0.0f;
    auto frag = [&](const char* p) {
        Split3 f;
        f.h = *reinterpret_cast<const u32x4*>(p);
        f.m = *reinterpret_cast<const u32x4*>(p + 1024);
        f.l = *reinterpret_cast<const u32x4*>(p + 2048);
        return f;
    };
    FA_STAMP(1);
    barrier();  // tile 0 is in stage 0
    FA_STAMP(2);
    for (int t = 0; t < nkt; ++t) {
        if (live) {
            const char* const kt = fa_lds + (t & 1) * FA_STAGE + lane * 16;
            const char* const vt = kt + FA_PART;
            // S^T tile: A = K (lane = key l31, k-slice hv), B = Q (lane = query l31, k-slice hv)
            // Two accumulators, alternating by chunk: an instruction between two MFMAs on the SAME accumulator stalls the matrix
            // core for ~43 cycles (the dependent pair must be adjacent), between different ones it costs its issue slot -- and
            // every chunk boundary carries the next fragment's three LDS reads, issued BEFORE the six products of the current
            // one (left to itself the scheduler sinks them to just above their first use).
            f32x16 st, st2;
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] = 0.0f, st2[r] = 0.0f;
            Split3 kf = frag(kt);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);  // (the groups below are matched in order: these three are chunk 0's)
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                Split3 kn = kf;
                if (c < 7) kn = frag(kt + (c + 1) * 3072);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                if (c & 1) mm6_32(kf, qs[c], st2);
                else mm6_32(kf, qs[c], st);
                __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                kf = kn;
            }
            Split3 vf = frag(vt);  // the first V^T fragment travels during the softmax
#pragma unroll
            for (int r = 0; r < 16; ++r) st[r] += st2[r];
            // online softmax: this lane holds query l31's scores for the keys 32 t + (r & 3) + 8 (r >> 2) + 4 hv; its partner lane the rest
            if (32 * t + 32 > a.tk) {  // only the last tile can hold keys that do not exist
#pragma unroll
                for (int r = 0; r < 16; ++r) st[r] = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hv < a.tk ? st[r] : -3.40282347e+38f;
            }
            float mt = st[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mt = fmaxf(mt, st[r]);
            mt = fmaxf(mt, swap32(mt));
            FA_STAMP(3 + 4 * t);  // the scores are out of the matrix core
            if (__builtin_amdgcn_ballot_w64(mt > m_run + 8.0f) != 0) {  // wave-uniform: some query's maximum outgrew its offset
                const float m_new = mt > m_run + 8.0f ? mt : m_run;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);  // 0 on the first tile (m_run = -FLT_MAX), 1 for a lane that stays
                l_run *= alpha;
                m_run = m_new;
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            float p[16], lsum = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                p[r] = __builtin_amdgcn_exp2f(st[r] - m_run);  // a masked key: exp2(-FLT_MAX - m) = 0
                lsum += p[r];
            }
            l_run += lsum;
            FA_STAMP(4 + 4 * t);
            // O^T += V^T P^T: B = P (lane = query, its 8 probabilities of key half c2 in register order), A = V^T (lane = output
            // dim, the same 8 keys in the same order: rows 16 c2 + 4 hv + {0..3} and 16 c2 + 8 + 4 hv + {0..3} of the tile)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) {
                const float pa[8] = {p[8 * c2], p[8 * c2 + 1], p[8 * c2 + 2], p[8 * c2 + 3], p[8 * c2 + 4], p[8 * c2 + 5], p[8 * c2 + 6], p[8 * c2 + 7]};
                const Split3 ps = split3(pa);
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    Split3 vn = vf;
                    if (c2 * 4 + d < 7) vn = frag(vt + (c2 * 4 + d + 1) * 3072);
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
                    mm6_32(vf, ps, o[d]);
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
                    vf = vn;
                }
            }
        }
        FA_STAMP(5 + 4 * t);
        barrier();  // done with this stage; the next tile is in the other one
        FA_STAMP(6 + 4 * t);
    }
    // every wave of the workgroup is past the last barrier: the stages are free.  O^T leaves through LDS so that the stores are
    // whole 512-byte rows (a lane owns a QUERY: stored directly, every instruction would touch 32 rows with 16 bytes each)
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
    if (live) {
        const float lfull = l_run + swap32(l_run);
        const float inv = 1.0f / lfull;
        constexpr int kRowB = 512 + 16;  // padded row: the 16-byte column writes of 32 rows spread over the banks
        char* const mine = fa_lds + wave * (32 * kRowB);
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float4 w;
                w.x = o[d][4 * g4] * inv, w.y = o[d][4 * g4 + 1] * inv, w.z = o[d][4 * g4 + 2] * inv, w.w = o[d][4 * g4 + 3] * inv;
                *reinterpret_cast<float4*>(mine + l31 * kRowB + (32 * d + 8 * g4 + 4 * hv) * 4) = w;
                if (rok) {
                    mn = fminf(mn, fminf(fminf(w.x, w.y), fminf(w.z, w.w)));
                    mx = fmaxf(mx, fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w)));
                }
            }
        float* obase = a.o + bo * a.o_so + bi * a.o_si + 4 * l31;
#pragma unroll
        for (int q = 0; q < 16; ++q) {  // rows 2 q + hv of the block: this wave wrote them itself (same-wave LDS order holds)
            const int r = 2 * q + hv;
            const float4 w = *reinterpret_cast<const float4*>(mine + r * kRowB + l31 * 16);
            if (i0 + r < a.tq) *reinterpret_cast<float4*>(obase + (int64_t)(i0 + r) * a.o_sr) = w;
        }
    }
    FA_STAMP(63);
    if (a.stat) {  // one {min, max} pair per compute wave (neutral for an idle one): 4 nqb pairs per (utterance, head)
        mn = wave_allreduce64(mn, [](float cur, float x) { return x < cur ? x : cur; });
        mx = wave_allreduce64(mx, [](float cur, float x) { return x > cur ? x : cur; });
        if (lane == 0) {
            float* dst = a.stat + (((int64_t)bo * a.batch_inner + bi) * (a.nqb * 4) + qb * 4 + wave) * 2;
            dst[0] = mn;
            dst[1] = mx;
        }
    }
}

bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

int lele_hip_attention_view(LeleCtx* ctx, const LeleTensor* q, const LeleMatView* qv, const LeleTensor* k, const LeleMatView* kv,
                            const LeleTensor* v, const LeleMatView* vv, int64_t batch_outer, int64_t batch_inner, int64_t t_q,
                            int64_t t_k, int64_t dh, const LeleTensor* scale, const LeleMatView* ov, const int64_t* out_dims,
                            int32_t out_dims_rank, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && q && qv && k && kv && v && vv && ov && out && out_dims, "attention_view: NULL argument");
    LELE_REQUIRE(q->dtype == LELE_F32 && k->dtype == LELE_F32 && v->dtype == LELE_F32, "attention_view: operands must be f32");
    LELE_REQUIRE(!scale || (scale->dtype == LELE_F32 && numel(scale) == 1), "attention_view: the scale must be one f32 value");
    LELE_REQUIRE(batch_outer >= 1 && batch_inner >= 1 && t_q >= 1 && t_k >= 1 && dh >= 1, "attention_view: bad dimensions");
    const int64_t fb = batch_outer * batch_inner;
    int64_t total = 1;
    for (int i = 0; i < out_dims_rank; ++i) total *= out_dims[i];
    LELE_REQUIRE(total == fb * t_q * dh, "attention_view: output shape holds %lld elements, the result has %lld", (long long)total,
                 (long long)(fb * t_q * dh));
    auto last = [&](const LeleMatView* w, int64_t r, int64_t c) {
        return w->offset + (batch_outer - 1) * w->stride_outer + (batch_inner - 1) * w->stride_inner + (r - 1) * w->stride_row + (c - 1) * w->stride_col;
    };
    // Q [t_q, dh], K^T [dh, t_k] (the B operand of Q K^T), V [t_k, dh], O [t_q, dh]
    LELE_REQUIRE(qv->offset >= 0 && last(qv, t_q, dh) < numel(q), "attention_view: the Q view leaves its tensor");
    LELE_REQUIRE(kv->offset >= 0 && last(kv, dh, t_k) < numel(k), "attention_view: the K view leaves its tensor");
    LELE_REQUIRE(vv->offset >= 0 && last(vv, t_k, dh) < numel(v), "attention_view: the V view leaves its tensor");
    LELE_REQUIRE(ov->offset >= 0 && last(ov, t_q, dh) < total, "attention_view: the output view leaves its buffer");
    // what the kernel handles; everything else is an error here -- the callers (lele_amd.plan / plan_runner.hpp) then issue
    // the three-call sequence this op stands for
    LELE_REQUIRE(dh == kDh && t_k <= 512 && qv->stride_col == 1 && kv->stride_row == 1 && vv->stride_col == 1 && ov->stride_col == 1,
                 "attention_view: unsupported geometry (head dimension %lld, %lld keys, unit strides %d%d%d%d)", (long long)dh, (long long)t_k,
                 (int)(qv->stride_col == 1), (int)(kv->stride_row == 1), (int)(vv->stride_col == 1), (int)(ov->stride_col == 1));
    LELE_REQUIRE(fb * ((t_q + 31) / 32) < (int64_t(1) << 31), "attention_view: too many blocks");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(ctx->arena_reset());
    const void *dq = nullptr, *dk = nullptr, *dv = nullptr, *dsc = nullptr;
    LELE_TRY(ctx->dev_ptr(q, &dq));
    LELE_TRY(ctx->dev_ptr(k, &dk));
    LELE_TRY(ctx->dev_ptr(v, &dv));
    if (scale) LELE_TRY(ctx->dev_ptr(scale, &dsc));
    LELE_TRY(out->reserve((size_t)total * 4));
    AttnArgs a{};
    a.q = (const float*)dq + qv->offset;
    a.k = (const float*)dk + kv->offset;
    a.v = (const float*)dv + vv->offset;
    a.o = (float*)out->data + ov->offset;
    a.q_so = qv->stride_outer, a.q_si = qv->stride_inner, a.q_sr = qv->stride_row;
    a.k_so = kv->stride_outer, a.k_si = kv->stride_inner, a.k_sr = kv->stride_col;  // K^T [dh, t_k]: a key is a COLUMN of the view
    a.v_so = vv->stride_outer, a.v_si = vv->stride_inner, a.v_sr = vv->stride_row;
    a.o_so = ov->stride_outer, a.o_si = ov->stride_inner, a.o_sr = ov->stride_row;
    // 16-byte fragment loads: every row start must be 16-byte aligned
    auto ok16 = [&](const float* p, int64_t so, int64_t si, int64_t sr) { return aligned16(p) && so % 4 == 0 && si % 4 == 0 && sr % 4 == 0; };
    LELE_REQUIRE(ok16(a.q, a.q_so, a.q_si, a.q_sr) && ok16(a.k, a.k_so, a.k_si, a.k_sr),
                 "attention_view: unsupported geometry (Q / K rows are not 16-byte aligned)");
    a.tq = (int)t_q, a.tk = (int)t_k, a.tpad = (int)((t_k + 63) & ~int64_t(63));
    a.batch_inner = (int)batch_inner;
    // two row tiles per workgroup when that still leaves three workgroups per CU (what fits at once); one otherwise (measured:
    // 32 x 171 rows 47.9 us with one tile against 54.2 us with two; 64 x 171 rows 92.6 against 73.8)
    // LELE_HIP_ATTENTION_EXACT=1: f32 MFMA products in the tiled GEMM's k order and the reference's row softmax -- the bits of the node
    // sequence (where that runs the tiled GEMM); default: split-bf16 products and the chip's exponential (same values to ~1e-6)
    const char* ex_env = getenv("LELE_HIP_ATTENTION_EXACT");
    const bool exact = ex_env && *ex_env && atoi(ex_env) != 0;
    const char* rt_env = lab_env("LELE_HIP_ATTENTION_RT");
    const int rt = rt_env && *rt_env ? atoi(rt_env) : (fb * ((t_q + 63) / 64) >= 3 * (int64_t)ctx->num_cus ? 2 : 1);
    // small grids (one utterance: 64 blocks of 32 rows for 256 CUs): 16 query rows per workgroup
    const char* rows_env = lab_env("LELE_HIP_ATTENTION_ROWS");
    const bool rows16 = rows_env && *rows_env ? atoi(rows_env) == 16 : fb * ((t_q + 31) / 32) < (int64_t)ctx->num_cus / 2;
    // one pass over the keys with producer waves (attention_flash_kernel): a batch of heads that gives at least half of the CUs a
    // workgroup of four 32-row blocks
    const bool flash = !exact && !rows16 && fb * ((t_q + FA_ROWS - 1) / FA_ROWS) >= (int64_t)ctx->num_cus / 2 && a.o_sr % 4 == 0 &&
                       aligned16(a.o) && a.o_so % 4 == 0 && a.o_si % 4 == 0 && !(rt_env && *rt_env) && !(rows_env && *rows_env);
    const int qrows = flash ? FA_ROWS : (rows16 ? 16 : (rt == 2 ? 64 : 32));
    a.nqb = (int)((t_q + qrows - 1) / qrows);
    a.scale = (const float*)dsc;
#ifdef LELE_HIP_LAB
    if (const char* e = lab_env("LELE_HIP_ATTN_ABLATE")) a.ablate = atoi(e);
    if (const char* e = lab_env("LELE_HIP_ATTN_STAMPS")) a.dbg = (long long*)(uintptr_t)strtoull(e, nullptr, 0);
#endif
    // result statistics for the dynamic quantisation that reads this tensor next: valid when a slice of the consumer is exactly
    // one outer batch element, i.e. the result is laid out [batch_outer][t_q][batch_inner * dh] (heads merged)
    const int64_t per_slice = (int64_t)a.batch_inner * a.nqb * (flash ? 4 : 1), nstat = batch_outer * per_slice;
    const bool merged = ov->stride_row == batch_inner * dh && ov->stride_inner == dh && ov->stride_outer == t_q * batch_inner * dh && ov->offset == 0;
    if (merged && nstat <= (int64_t(1) << 22)) {
        LELE_TRY(out->reserve_rowstat(nstat));
        if ((size_t)nstat <= out->rowstat_cap) a.stat = out->rowstat;
    }
    const size_t lds = (size_t)qrows * (a.tpad + kSPad) * 4;
    const dim3 grid((unsigned)(fb * a.nqb));
#define LELE_ATTN1(KERN_)                                                                                                  \
    do {                                                                                                                 \
        auto kern = KERN_;                                                                                               \
        if (lds > 60 * 1024) LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));               \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, a);                                                   \
    } while (0)
#define LELE_ATTN(NT_, RT_)                                              \
    do {                                                               \
        if (exact) LELE_ATTN1((attention_kernel<NT_, RT_, true>));      \
        else LELE_ATTN1((attention_kernel<NT_, RT_, false>));           \
    } while (0)
    // softmax registers per lane = key tiles exactly (tpad / 32, even): a 10 s utterance needs 6, not 8 -- the row softmax is a third
    // of the kernel's instructions (tools/attention_stamps.py) and every surplus register row is exponentials nobody reads
#define LELE_ATTN_NT(NT_)                                                                                      \
    do {                                                                                                       \
        if (rows16) { /* eight waves measured no better: 24.5 against 24.1 us */                               \
            if (exact) LELE_ATTN1((attention16_kernel<NT_, 4, true>));                                         \
            else LELE_ATTN1((attention16_kernel<NT_, 4, false>));                                              \
        } else if (rt == 2) LELE_ATTN(NT_, 2);                                                                 \
        else LELE_ATTN(NT_, 1);                                                                                \
    } while (0)
    if (flash) {
        auto kern = attention_flash_kernel;
        LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), FA_LDS));
        hipLaunchKernelGGL(kern, grid, dim3(512), FA_LDS, ctx->stream, a);
    } else
    switch (a.tpad / 64) {
        case 1: LELE_ATTN_NT(2); break;
        case 2: LELE_ATTN_NT(4); break;
        case 3: LELE_ATTN_NT(6); break;
        case 4: LELE_ATTN_NT(8); break;
        case 5: LELE_ATTN_NT(10); break;
        case 6: LELE_ATTN_NT(12); break;
        case 7: LELE_ATTN_NT(14); break;
        default: LELE_ATTN_NT(16); break;
    }
#undef LELE_ATTN_NT
#undef LELE_ATTN
#undef LELE_ATTN1
    LELE_HIP_CHECK(hipGetLastError());
    if (a.stat) {
        out->rowstat_rows = nstat;
        out->rowstat_len = batch_inner * dh;
        out->rowstat_m = t_q;
        out->rowstat_batch = batch_outer;
        out->rowstat_kind = 2;
        out->rowstat_valid = true;
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(out_dims, out_dims + out_dims_rank));
}

}  // extern "C"
