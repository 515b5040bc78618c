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
    int ablate;        // developer switch LELE_HIP_ATTN_ABLATE (timing experiments, results wrong): 1 no softmax arithmetic, 2 no exp
    long long* dbg;    // developer switch LELE_HIP_ATTN_STAMPS: 8 cycle-counter stamps per workgroup (tools/attention_stamps.py), or NULL
};

constexpr int kDh = 128;   // head dimension (8 chunks of 16)
constexpr int kSPad = 4;   // LDS row padding (floats)

// NT = softmax registers per lane (tpad <= 32 * NT).  RT = 32-row tiles of queries per workgroup (1 or 2).  With RT = 2 the waves
// pair up: wave w owns row tile w & 1 throughout; in phase 1 it takes every second key tile of that row tile (three each for the six
// tiles of a 10 s utterance: balanced, where RT = 1 leaves two of four waves with half the work), in phase 3 two of the four
// 32-dim output tiles, fed by ONE set of P fragments.  K and V are read half as often.
template <int NT, int RT>
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
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 8 + nstamp++] = (long long)clock64();
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
    auto mmk = [&](const float4 (&w)[8], int half, f32x16& acc) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 q0 = qf[2 * (4 * half + c)], q1 = qf[2 * (4 * half + c) + 1];
            const float qa[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            const float kk[8] = {w[2 * c].x, w[2 * c].y, w[2 * c].z, w[2 * c].w, w[2 * c + 1].x, w[2 * c + 1].y, w[2 * c + 1].z, w[2 * c + 1].w};
#pragma unroll
            for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[s], kk[s], acc, 0, 0, 0);
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
            for (int q = 0; q < 4; ++q)
                if (!(a.ablate & 1)) softmax_row_reg<NT>(v[q], a.tk, l);
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
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int d = 0; d < ND; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s], w[d][8 * c + s], oacc[d], 0, 0, 0);
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
template <int NT, int NW>
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
        if (a.dbg && tid == 0) a.dbg[(size_t)blockIdx.x * 8 + nstamp++] = (long long)clock64();
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
    auto mmk = [&](const float4 (&w)[8], int t) {
        f32x4v acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].x, w[c].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].y, w[c].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].z, w[c].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[c].w, w[c].w, acc, 0, 0, 0);
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
        for (int q = 0; q < RPG; ++q) softmax_row_reg<NT>(v[q], a.tk, l);
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
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const float4 p = *reinterpret_cast<const float4*>(prow + c0 + 4 * u4);
            const float pa[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int d = 0; d < ND; ++d) oacc[d] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[e], w[d][4 * u4 + e], oacc[d], 0, 0, 0);
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
    const char* rt_env = getenv("LELE_HIP_ATTENTION_RT");
    const int rt = rt_env && *rt_env ? atoi(rt_env) : (fb * ((t_q + 63) / 64) >= 3 * (int64_t)ctx->num_cus ? 2 : 1);
    // small grids (one utterance: 64 blocks of 32 rows for 256 CUs): 16 query rows per workgroup
    const char* rows_env = getenv("LELE_HIP_ATTENTION_ROWS");
    const bool rows16 = rows_env && *rows_env ? atoi(rows_env) == 16 : fb * ((t_q + 31) / 32) < (int64_t)ctx->num_cus / 2;
    const int qrows = rows16 ? 16 : (rt == 2 ? 64 : 32);
    a.nqb = (int)((t_q + qrows - 1) / qrows);
    a.scale = (const float*)dsc;
    a.dbg = nullptr;
    a.ablate = 0;
    if (const char* e = getenv("LELE_HIP_ATTN_ABLATE")) a.ablate = atoi(e);
    if (const char* e = getenv("LELE_HIP_ATTN_STAMPS")) a.dbg = (long long*)(uintptr_t)strtoull(e, nullptr, 0);
    // result statistics for the dynamic quantisation that reads this tensor next: valid when a slice of the consumer is exactly
    // one outer batch element, i.e. the result is laid out [batch_outer][t_q][batch_inner * dh] (heads merged)
    const int64_t per_slice = (int64_t)a.batch_inner * a.nqb, nstat = batch_outer * per_slice;
    const bool merged = ov->stride_row == batch_inner * dh && ov->stride_inner == dh && ov->stride_outer == t_q * batch_inner * dh && ov->offset == 0;
    if (merged && nstat <= (int64_t(1) << 22)) {
        LELE_TRY(out->reserve_rowstat(nstat));
        if ((size_t)nstat <= out->rowstat_cap) a.stat = out->rowstat;
    }
    const size_t lds = (size_t)qrows * (a.tpad + kSPad) * 4;
    const dim3 grid((unsigned)(fb * a.nqb));
#define LELE_ATTN(NT_, RT_)                                                                                                \
    do {                                                                                                                 \
        auto kern = attention_kernel<NT_, RT_>;                                                                           \
        if (lds > 60 * 1024) LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));               \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, a);                                                   \
    } while (0)
    // softmax registers per lane = key tiles exactly (tpad / 32, even): a 10 s utterance needs 6, not 8 -- the row softmax is a third
    // of the kernel's instructions (tools/attention_stamps.py) and every surplus register row is exponentials nobody reads
#define LELE_ATTN_NT(NT_)                                                                                      \
    do {                                                                                                       \
        if (rows16) {                                                                                          \
            auto kern = attention16_kernel<NT_, 4>; /* eight waves measured no better: 24.5 against 24.1 us */ \
            if (lds > 60 * 1024) LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds)); \
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, ctx->stream, a);                                     \
        } else if (rt == 2) LELE_ATTN(NT_, 2);                                                                 \
        else LELE_ATTN(NT_, 1);                                                                                \
    } while (0)
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
