// quant.hip -- lele's dynamic-quantised u8 linear path on gfx950 (i8 MFMA, v_mfma_i32_32x32x32_i8).
//
//   lele_hip_fused_quantized_linear     <- /root/reference/src/kernels/quantization.rs:77-169 (x86 branch) +
//                                          src/kernels/avx/quantization.rs:102-417 (DynQuant + u8 GEMM + epilogue)
//   lele_hip_dynamic_quantize_linear    <- quantization.rs:1628-1657 + avx/quantization.rs:832-927
//   lele_hip_mat_mul_integer_with_scale_bias (zero points, scale, bias, ReLU all optional)
//                                       <- quantization.rs:8-72, 927-992 + avx/quantization.rs:642-830
//
// Bit-exact with the reference: the integer part is exact, every rounding step is reproduced
// (round-half-even(fma(x,1/scale,zp)) in the 8-wide body, f32::round(x*inv+zp) in the per-row scalar tail; epilogue
// (float)acc * (dyn_scale*w_scale[j]) then + bias then max(.,0), mul and add NOT fused).
//
// Pipeline per call (all on the ctx stream):
//   1. qminmax_kernel + qparams_kernel : min/max of each batch slice -> {scale, zp, 1/scale}   (one range per slice!)
//   2. qrows_kernel                    : f32 rows -> i8 (q - 128) rows padded to 16, + exact i32 row sums
//   3. igemm_kernel                    : sum (q-128)(w-128) on the matrix cores; zero-point algebra + f32 epilogue
// Weights (f32-encoded u8 [K,N], as lele's TensorView::from_bytes_u8 carries them) are transposed once to i8
// [N, Kpad] (w - 128) with column sums and cached per (pointer, bytes) -- the analogue of B_WEIGHT_CACHE
// (avx/quantization.rs:12-95).
#include "common.h"
#include "lane_ops.h"
#include "gemm_small.h"
#include "norm_core.h"

#include <math.h>
#include <stdlib.h>

#include <algorithm>

using namespace lele;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

struct QParams {  // per batch slice
    float scale, zp, inv_scale;
    int zp_i;
};

// ------------------------------------------------------------------------------------------ 1. range
__global__ __launch_bounds__(256) void qminmax_kernel(const float* __restrict__ x, int64_t slice_len,
                                                      float* __restrict__ partial /*[slices][blocks][2]*/) {
    const float* p = x + (int64_t)blockIdx.y * slice_len;
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
    auto upd = [&](float v) {
        mn = v < mn ? v : mn;
        mx = v > mx ? v : mx;
    };
    // scalar head up to the first 16-byte boundary, float4 body, scalar tail (min/max are order-independent: exact)
    const int64_t head = std::min<int64_t>(slice_len, (int64_t)((4 - (((uintptr_t)p >> 2) & 3)) & 3));
    const int64_t nvec = (slice_len - head) / 4;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = (int64_t)gridDim.x * blockDim.x;
    if (gtid < head) upd(p[gtid]);
    const float4* pv = reinterpret_cast<const float4*>(p + head);
    for (int64_t i = gtid; i < nvec; i += gstride) {
        const float4 v = pv[i];
        upd(v.x);
        upd(v.y);
        upd(v.z);
        upd(v.w);
    }
    for (int64_t i = head + nvec * 4 + gtid; i < slice_len; i += gstride) upd(p[i]);
    for (int off = 32; off > 0; off >>= 1) {
        const float a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    __shared__ float smn[4], smx[4];
    if ((threadIdx.x & 63) == 0) {
        smn[threadIdx.x >> 6] = mn;
        smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) {
            mn = smn[w] < mn ? smn[w] : mn;
            mx = smx[w] > mx ? smx[w] : mx;
        }
        float* o = partial + ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 2;
        o[0] = mn;
        o[1] = mx;
    }
}

// Two instructions that do on their own what the reference spells out in several (probed on gfx950: profiles/r05_cvt_pk_u8_probe.txt):
//   * v_cvt_pk_u8_f32 rounds to nearest EVEN and saturates to [0, 255] (0.5 -> 0, 1.5 -> 2, 2.5 -> 2, 254.5 -> 254, 255.5 and
//     +inf -> 255, negative values and NaN -> 0): _mm256_round_ps(.., NEAREST) + the clamp + the pack in one -- no rintf in front;
//   * v_max_f32(x, +0) returns +0 for x = -0, for negative x and for NaN: `if x > 0 { x } else { 0 }` (ReLU) bit for bit.
__device__ __forceinline__ float relu0(float v) { return __builtin_fmaxf(v, 0.0f); }

// avx/quantization.rs:134-140: range -> {scale, zero point, 1/scale}
__device__ __forceinline__ QParams make_qparams(float mn, float mx) {
    const float adjusted_max = mx > 0.0f ? mx : 0.0f;
    const float adjusted_min = mn < 0.0f ? mn : 0.0f;
    float range = adjusted_max - adjusted_min;
    if (!(range > 1e-5f)) range = 1e-5f;
    const float scale = range / 255.0f;
    float z = roundf(-adjusted_min / scale);  // f32::round (half away from zero)
    z = z < 0.0f ? 0.0f : (z > 255.0f ? 255.0f : z);
    QParams q;
    q.scale = scale;
    q.zp = z;
    q.inv_scale = 1.0f / scale;
    q.zp_i = (int)z;
    return q;
}

__global__ void qparams_kernel(const float* __restrict__ partial, int nblocks, QParams* __restrict__ prm,
                               float* __restrict__ scale_out, float* __restrict__ zp_out, unsigned* __restrict__ zero_slice = nullptr) {
    const int s = blockIdx.x;
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
    for (int i = threadIdx.x; i < nblocks; i += 64) {
        const float a = partial[((int64_t)s * nblocks + i) * 2], b = partial[((int64_t)s * nblocks + i) * 2 + 1];
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float a = __shfl_xor(mn, off), b = __shfl_xor(mx, off);
        mn = a < mn ? a : mn;
        mx = b > mx ? b : mx;
    }
    if (threadIdx.x == 0) {
        const QParams q = make_qparams(mn, mx);
        prm[s] = q;
        if (scale_out) scale_out[s] = q.scale;
        if (zp_out) zp_out[s] = q.zp;
        if (zero_slice) zero_slice[s] = 0u;  // the per-slice maxima a feed-forward block's range pass adds into
    }
}

__device__ __forceinline__ float quant_one(float v, const QParams& q, bool simd_body) {
    float r = simd_body ? rintf(__builtin_fmaf(v, q.inv_scale, q.zp))  // _mm256_fmadd_ps + round-to-nearest-even
                        : roundf(v * q.inv_scale + q.zp);               // scalar remainder: two roundings, half away
    return r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
}

// ------------------------------------------------------------------------------------------ 2. rows -> i8
// MODE 0: dynamic quantisation with prm[row / m] (fused linear: SIMD body = first k&~7 elements of EACH ROW)
// MODE 1: the input already holds u8 values as f32 (mat_mul_integer): q = sat_u8(round-nearest-even(x))
// One wave per row; writes q-128 as i8 into [rows][kp] (zero padded = contributes 0) and the exact sum of (q-128).
// MODE 0 with `partial` != NULL: the slice's {scale, zp} are derived here from the min/max partials (every wave
// repeats the same few-hundred-element reduction, which is cheaper than a separate launch) and the first row of each
// slice publishes them to prm[] for the GEMM epilogue.
template <int MODE, bool PREFETCH = false>
__global__ __launch_bounds__(256) void qrows_kernel(const float* __restrict__ x, int64_t rows, int k, int kp, int m,
                                                    QParams* __restrict__ prm, int8_t* __restrict__ aq,
                                                    int* __restrict__ row_sums, const float* __restrict__ partial,
                                                    int nblk, unsigned* __restrict__ zero_slice = nullptr,
                                                    int* __restrict__ zero_rows = nullptr) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // side job for the fused two-layer form (lele_hip_fused_ffn_quantized): clear the accumulators its GEMM passes add into --
    // one per slice (the hidden layer's maximum) and one per row (the hidden layer's i8 row sums)
    if (lane == 0) {
        if (zero_rows) zero_rows[row] = 0;
        if (zero_slice && row % m == 0) zero_slice[row / m] = 0u;
    }
    // PREFETCH (few rows: the launch is latency-bound): the row's elements are requested FIRST (up to 8 x 16 bytes per lane,
    // K <= 2048), so that they travel while the range partials are fetched and reduced below -- two dependent round trips
    // become one.  With many rows the streaming loop is better (more waves in flight per SIMD): +3 % on the batched case.
    constexpr int MAXC = 8;
    const float* xrow = x + row * k;
    const bool fast = PREFETCH && (k & 3) == 0 && k >= 4 && (((uintptr_t)x & 15) == 0) && kp <= 256 * MAXC;
    const int nch = (kp + 255) / 256;
    float4 pre[MAXC];
    if (fast) {
#pragma unroll
        for (int u = 0; u < MAXC; ++u)
            if (u < nch) {  // uniform
                const int c = lane * 4 + 256 * u;
                pre[u] = *reinterpret_cast<const float4*>(xrow + (c < k ? c : k - 4));
            }
    }
    QParams q;
    if (MODE == 0) {
        const int64_t slice = row / m;
        if (partial) {
            float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
            // {min, max} pairs as 8-byte loads, all of a trip's loads in flight together (clamped index: a repeated pair changes
            // nothing); a one-pair-per-trip loop exposed one L2 round trip per 64 pairs -- 4 us for the 504 row pairs of a LayerNorm
            const float2* pp = reinterpret_cast<const float2*>(partial) + slice * nblk;
            auto sweep = [&](int i0, auto unroll) {
                constexpr int U = decltype(unroll)::value;
                float2 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = i0 + lane + 64 * u;
                    v[u] = pp[idx < nblk ? idx : nblk - 1];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    mn = v[u].x < mn ? v[u].x : mn;
                    mx = v[u].y > mx ? v[u].y : mx;
                }
            };
            if (nblk <= 256) {
                sweep(0, std::integral_constant<int, 4>());
            } else {  // up to 1024 pairs (a GEMM's per-workgroup pairs, a LayerNorm's rows) per trip: one round trip, not four
                for (int i0 = 0; i0 < nblk; i0 += 1024) sweep(i0, std::integral_constant<int, 16>());
            }
            // min / max over the wave on the DPP / permlane network (lane_ops.h; order-independent, exact)
            mn = wave_allreduce64(mn, [](float cur, float a) { return a < cur ? a : cur; });
            mx = wave_allreduce64(mx, [](float cur, float a) { return a > cur ? a : cur; });
            q = make_qparams(mn, mx);
            if (lane == 0 && row == slice * m) prm[slice] = q;
        } else {
            q = prm[slice];
        }
    }
    const float* xr = x + row * k;
    int8_t* dst = aq + row * kp;
    const int simd_k = k & ~7;
    const bool vec4 = (k & 3) == 0 && (((uintptr_t)x & 15) == 0);  // every row start is then 16-byte aligned
    int sum = 0;
    if (fast) {
        unsigned usum = 0;  // sum of q over the chunks quantised the short way
        int nfast = 0;
#pragma unroll
        for (int u = 0; u < MAXC; ++u)
            if (u < nch) {
                const int c = lane * 4 + 256 * u;
                if (MODE == 0 && c + 3 < simd_k) {
                    // four elements inside the SIMD body: fma, v_cvt_pk_u8_f32 (rounds to nearest even, saturates to [0, 255] = the
                    // clamp, and packs); per four elements one xor 0x80808080 (q - 128 as i8) and one v_sad_u8 (their sum)
                    unsigned pk = 0;
                    pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(pre[u].x, q.inv_scale, q.zp), 0, pk);
                    pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(pre[u].y, q.inv_scale, q.zp), 1, pk);
                    pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(pre[u].z, q.inv_scale, q.zp), 2, pk);
                    pk = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(pre[u].w, q.inv_scale, q.zp), 3, pk);
                    usum = __builtin_amdgcn_sad_u8(pk, 0u, usum);
                    nfast += 4;
                    *reinterpret_cast<unsigned*>(dst + c) = pk ^ 0x80808080u;
                } else if (c < kp) {
                    const float xv[4] = {pre[u].x, pre[u].y, pre[u].z, pre[u].w};
                    int packed = 0;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kk = c + e;
                        int v = 0;
                        if (kk < k) {
                            float qf;
                            if (MODE == 0)
                                qf = quant_one(xv[e], q, kk < simd_k);
                            else {
                                const float r = rintf(xv[e]);
                                qf = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
                            }
                            v = (int)qf - 128;
                            sum += v;
                        }
                        packed |= (v & 0xff) << (8 * e);
                    }
                    *reinterpret_cast<int*>(dst + c) = packed;
                }
            }
        sum += (int)usum - 128 * nfast;
    } else
    for (int c = lane * 4; c < kp; c += 256) {
        float xv[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (vec4 && c + 3 < k) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[e] = xr[c + e < k ? c + e : k - 1];
        }
        int packed = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int kk = c + e;
            int v = 0;
            if (kk < k) {
                float qf;
                if (MODE == 0)
                    qf = quant_one(xv[e], q, kk < simd_k);
                else {
                    const float r = rintf(xv[e]);
                    qf = r < 0.0f ? 0.0f : (r > 255.0f ? 255.0f : r);
                }
                v = (int)qf - 128;
                sum += v;
            }
            packed |= (v & 0xff) << (8 * e);
        }
        *reinterpret_cast<int*>(dst + c) = packed;
    }
    sum = wave_sum_i32(sum);
    if (lane == 0) row_sums[row] = sum;
}

// unfused DynamicQuantizeLinear: y (u8 values as f32), body = first len&~7 elements of the FLAT tensor
__global__ void dq_apply_kernel(const float* __restrict__ x, int64_t len, const QParams* __restrict__ prm,
                                float* __restrict__ y) {
    const QParams q = prm[0];
    const int64_t simd_end = (len / 8) * 8;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = quant_one(x[i], q, i < simd_end);
}

// ------------------------------------------------------------------------------------------ weights -> i8 [N][Kp]
// grid (64-column groups, 64-byte chunks of kp), 256 threads: a 64 x 64 tile is read coalesced along n (16 rows per thread, all
// loads in flight), centred, turned in LDS and written as 16-byte pieces of the [N][Kp] rows; column sums by atomicAdd on zeroed
// sums (a thread per column walking all of k took 240-410 us per matrix: ~100 ms of a SenseVoice-shaped model's first forward)
__global__ __launch_bounds__(256) void wpack_kernel(const float* __restrict__ w /*[K][N]*/, int k, int n, int kp, int8_t* __restrict__ wt,
                                                    int* __restrict__ col_sums) {
    __shared__ __attribute__((aligned(16))) int8_t s_t[64][80];  // [column][k], 16 bytes of padding per row
    __shared__ int s_part[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int j = j0 + c, jc = j < n ? j : n - 1;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int kk = k0 + q * 16 + e;
        v[e] = w[(int64_t)(kk < k ? kk : k - 1) * n + jc];
    }
    int sum = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        int val = 0;
        if (k0 + q * 16 + e < k) {
            const float r = rintf(v[e]);  // cvtps + saturating packs (transpose_b_from_f32_avx2)
            val = (r < 0.0f ? 0 : (r > 255.0f ? 255 : (int)r)) - 128;  // == u8 ^ 0x80 read as i8
            sum += val;
        }
        s_t[c][q * 16 + e] = (int8_t)val;
    }
    s_part[q][c] = sum;
    __syncthreads();
    if (q == 0 && j < n) atomicAdd(&col_sums[j], s_part[0][c] + s_part[1][c] + s_part[2][c] + s_part[3][c]);
    const int cc = threadIdx.x >> 2, seg = threadIdx.x & 3;
    if (j0 + cc < n && k0 + 16 * seg < kp)  // kp is a multiple of 16: a piece is whole or absent
        *reinterpret_cast<v4i*>(wt + (int64_t)(j0 + cc) * kp + k0 + 16 * seg) = *reinterpret_cast<const v4i*>(&s_t[cc][16 * seg]);
}

// ------------------------------------------------------------------------------------------ 3. i8 GEMM
// C[row][col] = sum_k A'[row][k] * B'[col][k]   (A' = q-128, B' = w-128, both [rows][kp] with k contiguous)
// Tile BM x BN, K step 64 bytes; LDS rows of 64 B + 16 B pad (same bank argument as the f32 core: 5*row mod 16).
// Lane l feeds the two MFMA steps of a K tile with bytes [32*(l>>5), +16) and [32*(l>>5)+16, +16) of its row, for A
// and B alike, so the hardware's internal k order is irrelevant (it pairs like with like).
struct IgemmEpi {
    float* out;
    int64_t rows, n;
    int m;  // rows per batch slice
    int k;
    const int* row_sums;  // sum_k (q-128) per row
    const int* col_sums;  // sum_k (w-128) per column
    const QParams* prm;   // per slice (dynamic) or NULL
    int zp_a_fixed, zp_b;
    const float* wscale;  // may be NULL
    int wscale_len;
    const float* bias;  // may be NULL
    int relu;
    // residual operands [rows][n] added after bias / ReLU, in this order: ((lin + res1) + res2) -- the Add nodes that follow
    // a projection in a transformer block, folded into its store (lele_hip_fused_quantized_linear_residual)
    const float* res1 = nullptr;
    const float* res2 = nullptr;
    float* blockstat = nullptr;  // small-problem kernel: one {min, max} pair per workgroup (common.h, LeleBuf::rowstat)
    int flags = 0;               // developer A/B switches (LELE_HIP_IGEMM_FLAGS): 1 = column terms after the K loop, 2 = previous epilogue
    // the fused two-layer form (igemm_kernel's EM = 1 / 2 passes, see lele_hip_fused_ffn_quantized)
    unsigned* slice_max = nullptr;  // per slice: bits of max(result) (ReLU results are >= 0: unsigned order = float order)
    int8_t* q_out = nullptr;        // EM 2: the result quantised with its slice's range, as q - 128, [rows][n]
    int* q_rowsum = nullptr;        // EM 2: sum over the row of (q - 128), accumulated over column blocks
    QParams* q_prm = nullptr;       // EM 2: the slices' parameters, published for the next GEMM's epilogue
    // Everything that depends only on the row (slice parameters, row-sum term, output row pointer) or only on the
    // column (column sum, weight scale, bias) is computed once per row / column of a thread's tile, not per element.
    struct RowCtx {
        int ca, rterm;
        float dyn_scale;
        float* orow;
    };
    struct ColCtx {
        int colsum;
        float ws, bias;
    };
    __device__ __forceinline__ RowCtx row_ctx(int64_t row) const {
        int zp_a = zp_a_fixed;
        float ds = 1.0f;
        if (prm) {
            const QParams q = prm[rows == m ? 0u : (unsigned)row / (unsigned)m];
            zp_a = q.zp_i;
            ds = q.scale;
        }
        // sum (q - zp_a)(w - zp_b) with q = q'+128, w = w'+128  (exact in i32, as the reference's wrapping algebra)
        const int ca = 128 - zp_a, cb = 128 - zp_b;
        return RowCtx{ca, cb * row_sums[row] + k * ca * cb, ds, out + row * n};
    }
    __device__ __forceinline__ ColCtx col_ctx(int col) const {
        ColCtx c{0, 1.0f, 0.0f};
        col = col < n ? col : (int)n - 1;  // clamped: loads stay unconditional, out-of-range columns are never stored
        c.colsum = col_sums[col];
        if (wscale) c.ws = wscale_len <= 1 ? wscale[0] : wscale[col];
        if (bias) c.bias = bias[col];
        return c;
    }
    __device__ __forceinline__ float value(const RowCtx& r, const ColCtx& c, int acc) const {
        const int total = acc + r.rterm + r.ca * c.colsum;
        float vf = (float)total;  // _mm256_cvtepi32_ps
        // combined_scale[j] = dyn_scale * weight_scale[j], then one mul (dyn_scale is 1.0 when there is no dynamic range)
        if (wscale) vf = vf * (r.dyn_scale * c.ws);
        if (bias) vf = vf + c.bias;
        if (relu) vf = relu0(vf);
        return vf;
    }
    // the same value with the zero-point product on the 24-bit multiplier (full rate): |128 - zp_a| <= 128 and a column sum of
    // at most 2^23 (K <= 65536) are exact there
    __device__ __forceinline__ float value24(const RowCtx& r, const ColCtx& c, int acc) const {
        const int total = acc + r.rterm + __mul24(r.ca, c.colsum);
        float vf = (float)total;
        if (wscale) vf = vf * (r.dyn_scale * c.ws);
        if (bias) vf = vf + c.bias;
        if (relu) vf = relu0(vf);
        return vf;
    }
    __device__ __forceinline__ void store(const RowCtx& r, const ColCtx& c, int col, int acc) const { r.orow[col] = value(r, c, acc); }
    // with residuals: the caller has loaded them (unconditionally, from clamped coordinates) before any store
    __device__ __forceinline__ float value_res(const RowCtx& r, const ColCtx& c, int acc, float r1, float r2) const {
        float vf = value(r, c, acc) + r1;
        if (res2) vf = vf + r2;
        return vf;
    }
    __device__ __forceinline__ void store_res(const RowCtx& r, const ColCtx& c, int col, int acc, float r1, float r2) const {
        r.orow[col] = value_res(r, c, acc, r1, r2);
    }
};

// EM (epilogue mode) 0: the result as f32.  EM 1 / 2 are the two passes of the fused two-layer form, where the f32 result (the
// hidden layer of a feed-forward block, 45 MB per configs[3] shard) never exists in HBM -- recomputing the product costs ~5 us of
// matrix-core time against ~30 us of traffic for writing it and reading it twice:
//   EM 1: only the per-slice maximum of the (ReLU) result, by atomic max into epi.slice_max;
//   EM 2: the result again, quantised with the slice's now-known range exactly as qrows_kernel would quantise the stored f32
//         (SIMD body: rint(fma(v, 1/scale, zp)), saturated), written as i8 with its row sums -- what the next GEMM consumes.
template <int BM, int BN, int WM, int WN, int BKB /* bytes of K per LDS tile: 64 or 128 */, int OCC = 1, int EM = 0>
__global__ __launch_bounds__(WM* WN * 64) __attribute__((amdgpu_waves_per_eu(OCC))) void igemm_kernel(const int8_t* __restrict__ a, const int8_t* __restrict__ b,
                                                            int64_t rows, int n, int kp, int64_t b_batch_stride,
                                                            int m_per_batch, IgemmEpi epi) {
    constexpr int NT = WM * WN * 64;
    constexpr int PITCH = BKB + 16;  // bytes (80 / 144: 16-B aligned rows, 16 distinct bank slots per 16-lane group)
    constexpr int CQ = BKB / 16;     // 16-B chunks per tile row
    constexpr int TMT = BM / WM / 32, TNT = BN / WN / 32;
    constexpr int ASLOTS = (BM * CQ + NT - 1) / NT, BSLOTS = (BN * CQ + NT - 1) / NT;  // 16-B chunks per thread
    extern __shared__ __attribute__((aligned(16))) char igemm_lds[];  // As[2][BM*PITCH] then Bs[2][BN*PITCH]
    char(*As)[BM * PITCH] = reinterpret_cast<char(*)[BM * PITCH]>(igemm_lds);
    char(*Bs)[BN * PITCH] = reinterpret_cast<char(*)[BN * PITCH]>(igemm_lds + 2 * BM * PITCH);
    __shared__ int s_ca[BM], s_rterm[BM];
    __shared__ float s_ds[BM];
    __shared__ unsigned s_mx[2];
    __shared__ QParams s_q[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    unsigned tx, ty, tz;
    gemm::tile_coords(tx, ty, tz);  // XCD-aware order (gemm_core.h)
    const int64_t m0 = (int64_t)ty * BM;
    // EM 1 / 2 (m >= BM): the tile's rows below `split` belong to slice s0, the rest to slice s0 + 1
    const int s0 = EM ? (int)(m0 / epi.m) : 0;
    const int split = EM ? (int)(((int64_t)s0 + 1) * epi.m - m0) : 0;
    if (EM == 1 && tid < 2) s_mx[tid] = 0u;
    if (EM == 2 && tid < 2) {
        const int64_t nslices = epi.rows / epi.m;
        const int64_t sl = s0 + tid < nslices ? s0 + tid : nslices - 1;
        s_q[tid] = make_qparams(0.0f, __uint_as_float(epi.slice_max[sl]));
    }
    // per-row epilogue terms: fetched once per block up front (clamped, unconditional -- their latency hides behind
    // the K loop) instead of per element behind a bounds branch, which serialises one global load per row
    for (int t = tid; t < BM; t += NT) {
        const int64_t r = m0 + t < rows ? m0 + t : rows - 1;
        const IgemmEpi::RowCtx rc = epi.row_ctx(r);
        s_ca[t] = rc.ca;
        s_rterm[t] = rc.rterm;
        s_ds[t] = rc.dyn_scale;
    }
    const int n0 = tx * BN;
    const int hv = lane >> 5, l31 = lane & 31;
    // un-batched weights: b_batch_stride == 0; batched B (mat_mul_integer with batch_b > 1): slice = row block / m
    const int8_t* bb = b + (b_batch_stride ? (m0 / m_per_batch) * b_batch_stride : 0);

    v4i ra[ASLOTS], rb[BSLOTS];
    const v4i zero4 = {0, 0, 0, 0};
    auto gload = [&](int k0) {
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            const int s = tid + i * NT, row = s / CQ, kq = s % CQ;
            if (s < BM * CQ) {
                const int64_t r = m0 + row;
                const int kb = k0 + 16 * kq;
                ra[i] = (r < rows && kb < kp) ? *reinterpret_cast<const v4i*>(a + r * kp + kb) : zero4;
            }
        }
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i) {
            const int s = tid + i * NT, row = s / CQ, kq = s % CQ;
            if (s < BN * CQ) {
                const int c = n0 + row;
                const int kb = k0 + 16 * kq;
                rb[i] = (c < n && kb < kp) ? *reinterpret_cast<const v4i*>(bb + (int64_t)c * kp + kb) : zero4;
            }
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < ASLOTS; ++i) {
            const int s = tid + i * NT;
            if (s < BM * CQ) *reinterpret_cast<v4i*>(&As[buf][(s / CQ) * PITCH + 16 * (s % CQ)]) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < BSLOTS; ++i) {
            const int s = tid + i * NT;
            if (s < BN * CQ) *reinterpret_cast<v4i*>(&Bs[buf][(s / CQ) * PITCH + 16 * (s % CQ)]) = rb[i];
        }
    };
    v16i acc[TMT][TNT];
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int j = 0; j < TNT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0;

    // the tiles' column terms (column sum, weight scale, bias): requested before the K loop and pinned there -- fetched after it
    // every block would wait one full memory round trip with nothing else to do
    IgemmEpi::ColCtx cc[TNT];
    int cols[TNT];
#pragma unroll
    for (int j = 0; j < TNT; ++j) cols[j] = n0 + wn * TNT * 32 + j * 32 + l31;
    if (!(epi.flags & 1)) {
#pragma unroll
        for (int j = 0; j < TNT; ++j) cc[j] = epi.col_ctx(cols[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int nk = (kp + BKB - 1) / BKB;
    gload(0);
    lstore(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * BKB);
#pragma unroll
        for (int sub = 0; sub < BKB / 64; ++sub) {  // 64 bytes of K per sub-step (two MFMA steps)
            v4i fa[TMT][2], fb[TNT][2];
#pragma unroll
            for (int i = 0; i < TMT; ++i) {
                const char* src = &As[cur][(wm * TMT * 32 + i * 32 + l31) * PITCH + sub * 64 + 32 * hv];
                fa[i][0] = *reinterpret_cast<const v4i*>(src);
                fa[i][1] = *reinterpret_cast<const v4i*>(src + 16);
            }
#pragma unroll
            for (int j = 0; j < TNT; ++j) {
                const char* src = &Bs[cur][(wn * TNT * 32 + j * 32 + l31) * PITCH + sub * 64 + 32 * hv];
                fb[j][0] = *reinterpret_cast<const v4i*>(src);
                fb[j][1] = *reinterpret_cast<const v4i*>(src + 16);
            }
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < TMT; ++i)
#pragma unroll
                    for (int j = 0; j < TNT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }
    if (epi.flags & 1) {
#pragma unroll
        for (int j = 0; j < TNT; ++j) cc[j] = epi.col_ctx(cols[j]);
    }
    if constexpr (EM == 1) {
        const int rows_here = (int)(rows - m0 < BM ? rows - m0 : BM);
        float mx0 = 0.0f, mx1 = 0.0f;  // ReLU: every value is >= 0, NaN becomes 0
#pragma unroll
        for (int i = 0; i < TMT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * TMT * 32 + i * 32 + 4 * hv + (r & 3) + 8 * (r >> 2);
                const IgemmEpi::RowCtx rc{s_ca[lr], s_rterm[lr], s_ds[lr], nullptr};
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    float v = epi.value24(rc, cc[j], acc[i][j][r]);
                    v = (lr < rows_here && cols[j] < n) ? v : 0.0f;
                    if (lr < split) mx0 = fmaxf(mx0, v);
                    else mx1 = fmaxf(mx1, v);
                }
            }
        for (int off = 32; off > 0; off >>= 1) {
            mx0 = fmaxf(mx0, __shfl_xor(mx0, off));
            mx1 = fmaxf(mx1, __shfl_xor(mx1, off));
        }
        if (lane == 0) {
            if (mx0 > 0.0f) atomicMax(&s_mx[0], __float_as_uint(mx0));
            if (mx1 > 0.0f) atomicMax(&s_mx[1], __float_as_uint(mx1));
        }
        __syncthreads();
        if (tid < 2 && s_mx[tid]) atomicMax(&epi.slice_max[s0 + tid], s_mx[tid]);
        return;
    }
    if constexpr (EM == 2) {
        static_assert(EM != 2 || BN % 16 == 0, "i8 tile rows are stored in 16-byte chunks");
        const int rows_here = (int)(rows - m0 < BM ? rows - m0 : BM);
        constexpr int QP = BN + 16;   // i8 tile [BM][BN] in the K loop's LDS (its last barrier has passed), rows padded
        char* const qt = igemm_lds;
#pragma unroll
        for (int i = 0; i < TMT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = wm * TMT * 32 + i * 32 + 4 * hv + (r & 3) + 8 * (r >> 2);
                const IgemmEpi::RowCtx rc{s_ca[lr], s_rterm[lr], s_ds[lr], nullptr};
                const int which = lr < split ? 0 : 1;
                const float inv = s_q[which].inv_scale, zp = s_q[which].zp;
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    const float v = epi.value24(rc, cc[j], acc[i][j][r]);
                    const unsigned q = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v, inv, zp), 0, 0u);
                    qt[lr * QP + wn * TNT * 32 + j * 32 + l31] = (char)(q ^ 0x80u);
                }
            }
        __syncthreads();
        constexpr int CH = BN / 16;  // 16-byte chunks per tile row: CH consecutive lanes hold one row
        static_assert(EM != 2 || (CH == 8 || CH == 4 || CH == 16), "row-sum reduction over CH lanes");
        for (int c = tid; c < BM * CH; c += NT) {
            const int row = c / CH, c16 = c % CH;
            const v4i w = *reinterpret_cast<const v4i*>(qt + row * QP + 16 * c16);
            unsigned us = __builtin_amdgcn_sad_u8((unsigned)w[0] ^ 0x80808080u, 0u, 0u);
            us = __builtin_amdgcn_sad_u8((unsigned)w[1] ^ 0x80808080u, 0u, us);
            us = __builtin_amdgcn_sad_u8((unsigned)w[2] ^ 0x80808080u, 0u, us);
            us = __builtin_amdgcn_sad_u8((unsigned)w[3] ^ 0x80808080u, 0u, us);
            int part = (int)us - 128 * 16;  // sum of (q - 128) over the chunk
            for (int off = CH / 2; off > 0; off >>= 1) part += __shfl_xor(part, off);
            if (row < rows_here) {
                *reinterpret_cast<v4i*>(epi.q_out + (m0 + row) * (int64_t)n + n0 + 16 * c16) = w;
                if (c16 == 0) atomicAdd(&epi.q_rowsum[m0 + row], part);
            }
        }
        if (tx == 0 && tid < 2 && (tid == 0 || split < rows_here)) epi.q_prm[s0 + tid] = s_q[tid];
        return;
    }
    if (epi.flags & 2) {
    if (epi.res1) {  // residual operands: four rows at a time, their loads issued together before the stores
#pragma unroll
        for (int i = 0; i < TMT; ++i)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float r1[4][TNT], r2[4][TNT];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int64_t row = m0 + wm * TMT * 32 + i * 32 + q + 8 * g4 + 4 * hv;
                    const int64_t rowc = row < rows ? row : rows - 1;
#pragma unroll
                    for (int j = 0; j < TNT; ++j) {
                        const int64_t at = rowc * epi.n + (cols[j] < n ? cols[j] : n - 1);
                        r1[q][j] = epi.res1[at];
                        r2[q][j] = epi.res2 ? epi.res2[at] : 0.0f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int lr = wm * TMT * 32 + i * 32 + q + 8 * g4 + 4 * hv;
                    const int64_t row = m0 + lr;
                    const IgemmEpi::RowCtx rc{s_ca[lr], s_rterm[lr], s_ds[lr], epi.out + (row < rows ? row : 0) * epi.n};
#pragma unroll
                    for (int j = 0; j < TNT; ++j)
                        if (row < rows && cols[j] < n) epi.store_res(rc, cc[j], cols[j], acc[i][j][q + 4 * g4], r1[q][j], r2[q][j]);
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TMT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int lr = wm * TMT * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hv;
            const int64_t row = m0 + lr;
            if (row >= rows) continue;
            const IgemmEpi::RowCtx rc{s_ca[lr], s_rterm[lr], s_ds[lr], epi.out + row * epi.n};
#pragma unroll
            for (int j = 0; j < TNT; ++j)
                if (cols[j] < n) epi.store(rc, cc[j], cols[j], acc[i][j][r]);
        }
        return;
    }
    // epilogue: straight-line per tile -- one uniform 64-bit base + 32-bit element offsets, the zero-point product on the 24-bit
    // multiplier, every residual operand of a tile requested before its first store (loads and stores retire through ONE
    // in-order counter: a load issued after a store cannot be waited for without waiting for the store)
    const int rows_here = (int)(rows - m0 < BM ? rows - m0 : BM);
    float* const obase = epi.out + m0 * (int64_t)n;
    const unsigned nu = (unsigned)n;
    auto epilogue = [&](auto nres_c) {
        constexpr int NRES = decltype(nres_c)::value;
        const float* const r1base = NRES > 0 ? epi.res1 + m0 * (int64_t)n : nullptr;
        const float* const r2base = NRES > 1 ? epi.res2 + m0 * (int64_t)n : nullptr;
#pragma unroll
        for (int i = 0; i < TMT; ++i) {
            const int rbase = wm * TMT * 32 + i * 32 + 4 * hv;
            float r1[NRES > 0 ? 16 : 1][TNT], r2[NRES > 1 ? 16 : 1][TNT];
            if (NRES > 0) {
                const int last = rows_here - 1;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int lr = rbase + (r & 3) + 8 * (r >> 2);
                    const unsigned rowoff = (unsigned)(lr < last ? lr : last) * nu;
#pragma unroll
                    for (int j = 0; j < TNT; ++j) {
                        const unsigned at = rowoff + (unsigned)(cols[j] < n ? cols[j] : n - 1);
                        r1[r][j] = r1base[at];
                        if (NRES > 1) r2[r][j] = r2base[at];
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int lr = rbase + (r & 3) + 8 * (r >> 2);
                const IgemmEpi::RowCtx rc{s_ca[lr], s_rterm[lr], s_ds[lr], nullptr};
                const bool rok = lr < rows_here;
                const unsigned rowoff = (unsigned)lr * nu;
#pragma unroll
                for (int j = 0; j < TNT; ++j) {
                    float v = epi.value24(rc, cc[j], acc[i][j][r]);
                    if (NRES > 0) v = v + r1[r][j];
                    if (NRES > 1) v = v + r2[r][j];
                    if (rok && cols[j] < n) obase[rowoff + (unsigned)cols[j]] = v;
                }
            }
        }
    };
    if (!epi.res1) epilogue(std::integral_constant<int, 0>());
    else if (!epi.res2) epilogue(std::integral_constant<int, 1>());
    else epilogue(std::integral_constant<int, 2>());
}


}  // namespace
#include "igemm_rs.h"
#include "igemm_big.h"
namespace {

// ------------------------------------------------------------------------------------------ fragment-major weights (igemm_rs.h)
// weights [K][N] (u8 values as f32) -> fragment-major i8: block (ct, ks) = columns [32ct, +32) x k [32ks, +32), lane l of the
// block holds column 32ct + (l & 31), k = 32ks + 16(l >> 5) + [0, 16).  One thread per (column, 16-k chunk).
__global__ void wpack_frag_kernel(const float* __restrict__ w, int k, int n, int ks, int nt, int8_t* __restrict__ wf) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int j = (int)(t % (nt * 32));          // column (fast: the reads of a wave are one coalesced row segment)
    const int kc = (int)(t / (nt * 32));         // 16-k chunk
    if (kc >= ks * 2) return;
    int packed[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int kk = kc * 16 + e;
        int v = 0;
        if (kk < k && j < n) {
            const float r = rintf(w[(int64_t)kk * n + j]);  // cvtps + saturating packs (transpose_b_from_f32_avx2)
            const int u = r < 0.0f ? 0 : (r > 255.0f ? 255 : (int)r);
            v = u - 128;
        }
        packed[e >> 2] |= (v & 0xff) << (8 * (e & 3));
    }
    const int64_t blk = (int64_t)(j >> 5) * ks + (kc >> 1);
    const int lane = (j & 31) + 32 * (kc & 1);
    v4i o = {packed[0], packed[1], packed[2], packed[3]};
    *reinterpret_cast<v4i*>(wf + (blk * 64 + lane) * 16) = o;
}
// column sums of the centred weights (exact i32): grid (64-column groups, 64-row chunks of k), 256 threads = 4 k-quarters x 64
// columns, 16 rows each with every load in flight at once; the quarters meet in LDS, the chunks by atomicAdd on zeroed sums
// (a thread per column walking all of k took 260 us per weight matrix: 54 ms of a SenseVoice-shaped model's first forward)
__global__ __launch_bounds__(256) void wcolsum_kernel(const float* __restrict__ w, int k, int n, int* __restrict__ col_sums) {
    __shared__ int s_part[4][64];
    const int c = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + c, jc = j < n ? j : n - 1;
    const int k0 = blockIdx.y * 64 + q * 16;
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int kk = k0 + e;
        v[e] = w[(int64_t)(kk < k ? kk : k - 1) * n + jc];
    }
    int sum = 0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float r = rintf(v[e]);
        sum += k0 + e < k ? (r < 0.0f ? 0 : (r > 255.0f ? 255 : (int)r)) - 128 : 0;
    }
    s_part[q][c] = sum;
    __syncthreads();
    if (q == 0 && j < n) atomicAdd(&col_sums[j], s_part[0][c] + s_part[1][c] + s_part[2][c] + s_part[3][c]);
}
int launch_wcolsum(LeleCtx* ctx, const float* dw, int k, int n, int* dcs) {
    LELE_HIP_CHECK(hipMemsetAsync(dcs, 0, (size_t)n * 4, ctx->stream));
    hipLaunchKernelGGL(wcolsum_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((k + 63) / 64)), dim3(256), 0, ctx->stream, dw, k, n, dcs);
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------ host helpers
struct PackedW {
    int8_t* wt;
    int* col_sums;
};

int get_packed_weights(LeleCtx* ctx, const LeleTensor* w, const float* dw, int64_t nbatch, int k, int n, int kp,
                       PackedW* out, bool cacheable) {
    // cache key: the caller's pointer (host pointer for LELE_MEM_WEIGHT, device pointer otherwise) + size
    auto key_w = std::make_tuple((const void*)w->data, (size_t)numel(w) * 4, 201);
    auto key_s = std::make_tuple((const void*)w->data, (size_t)numel(w) * 4, 202);
    if (cacheable) {
        auto it = ctx->weights.find(key_w);
        if (it != ctx->weights.end()) {
            out->wt = (int8_t*)it->second;
            out->col_sums = (int*)ctx->weights[key_s];
            return 0;
        }
    }
    void *dwt = nullptr, *dcs = nullptr;
    if (cacheable) {
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMalloc(&dwt, (size_t)nbatch * n * kp));
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMalloc(&dcs, (size_t)nbatch * n * 4));
        ctx->weights[key_w] = dwt;
        ctx->weights[key_s] = dcs;
    } else {
        LELE_TRY(ctx->arena_alloc((size_t)nbatch * n * kp, &dwt));
        LELE_TRY(ctx->arena_alloc((size_t)nbatch * n * 4, &dcs));
    }
    LELE_HIP_CHECK(hipMemsetAsync(dcs, 0, (size_t)nbatch * n * 4, ctx->stream));
    for (int64_t b = 0; b < nbatch; ++b)
        hipLaunchKernelGGL(wpack_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((kp + 63) / 64)), dim3(256), 0, ctx->stream,
                           dw + b * (int64_t)k * n, k, n, kp, (int8_t*)dwt + b * (int64_t)n * kp, (int*)dcs + b * n);
    LELE_HIP_CHECK(hipGetLastError());
    out->wt = (int8_t*)dwt;
    out->col_sums = (int*)dcs;
    return 0;
}


// fragment-major packed weights for the one-pass kernel (cached like PackedW; tags 203 / 204)
struct FragW {
    int8_t* wf;
    int* col_sums;
};
int get_frag_weights(LeleCtx* ctx, const LeleTensor* w, int k, int n, int ks, int nt, FragW* out) {
    auto key_w = std::make_tuple((const void*)w->data, (size_t)numel(w) * 4, 203);
    auto key_s = std::make_tuple((const void*)w->data, (size_t)numel(w) * 4, 204);
    auto it = ctx->weights.find(key_w);
    if (it != ctx->weights.end()) {
        out->wf = (int8_t*)it->second;
        out->col_sums = (int*)ctx->weights[key_s];
        return 0;
    }
    LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
    const void* dw = nullptr;
    LELE_TRY(ctx->dev_ptr(w, &dw));  // the f32-encoded matrix (uploaded and cached under tag 0)
    void *dwf = nullptr, *dcs = nullptr;
    LELE_HIP_CHECK(hipMalloc(&dwf, (size_t)nt * ks * 1024));
    LELE_HIP_CHECK(hipMalloc(&dcs, (size_t)n * 4));
    const int64_t threads = (int64_t)nt * 32 * ks * 2;
    hipLaunchKernelGGL(wpack_frag_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)dw, k, n, ks,
                       nt, (int8_t*)dwf);
    LELE_TRY(launch_wcolsum(ctx, (const float*)dw, k, n, (int*)dcs));
    LELE_HIP_CHECK(hipGetLastError());
    ctx->weights[key_w] = dwf;
    ctx->weights[key_s] = dcs;
    out->wf = (int8_t*)dwf;
    out->col_sums = (int*)dcs;
    return 0;
}

int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return v && *v ? atoi(v) : dflt;
}
// developer switches exist in the lab build only (LELE_HIP_LAB=1 python -m lele_amd.build); the product reads the documented ones
int lab_int(const char* name, int dflt) {
#ifdef LELE_HIP_LAB
    return env_int(name, dflt);
#else
    (void)name;
    return dflt;
#endif
}

// stage stopwatch of the quantised linear (lele_hip_quant_set_profiling): stage = 0 start, 1 after the range pass, 2 after the
// row quantisation, 3 after the GEMM
int qprof_mark(LeleCtx* ctx, int stage) {
    if (!ctx->qprof.on || ctx->capturing) return 0;
    auto& p = ctx->qprof;
    if (stage == 0) {
        if (p.used + 4 > p.ev.size()) {
            for (int i = 0; i < 4; ++i) {
                hipEvent_t e;
                LELE_HIP_CHECK(hipEventCreate(&e));
                p.ev.push_back(e);
            }
        }
        p.used += 4;
    }
    LELE_HIP_CHECK(hipEventRecord(p.ev[p.used - 4 + stage], ctx->stream));
    return 0;
}

// fused != NULL: only the min/max partials are produced (at most 256 per slice); *fused / *fused_nblk receive them and the
// caller's qrows_kernel<0> turns them into parameters -- one launch fewer on the latency-bound path.
int launch_range(LeleCtx* ctx, const float* dx, int64_t slices, int64_t slice_len, QParams* prm, float* scale_out,
                 float* zp_out, const float** fused = nullptr, int* fused_nblk = nullptr) {
    // enough blocks to fill 256 CUs x 4 even for one slice; every thread then streams >= 4 float4
    const int64_t want = (slice_len + 4095) / 4096;
    const int nblk = (int)std::max<int64_t>(1, std::min<int64_t>(std::max<int64_t>(1, 1024 / slices), want));  // <= 1024 partial pairs in total; a fused reader scans its slice's (<= 1024) pairs
    void* partial = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)slices * nblk * 8, &partial));
    hipLaunchKernelGGL(qminmax_kernel, dim3(nblk, (unsigned)slices), dim3(256), 0, ctx->stream, dx, slice_len,
                       (float*)partial);
    if (fused) {
        *fused = (const float*)partial;
        *fused_nblk = nblk;
    } else {
        hipLaunchKernelGGL(qparams_kernel, dim3((unsigned)slices), dim3(64), 0, ctx->stream, (const float*)partial, nblk,
                           prm, scale_out, zp_out);
    }
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_igemm(LeleCtx* ctx, const int8_t* aq, const int8_t* wt, int64_t rows, int n, int kp, int64_t b_stride,
                 int m_per_batch, const IgemmEpi& epi_in) {
    if (rows == 0 || n == 0) return 0;
    IgemmEpi epi = epi_in;
    epi.flags = lab_int("LELE_HIP_IGEMM_FLAGS", 0);
    LELE_REQUIRE(rows < (int64_t(1) << 31), "quantized GEMM: more than 2^31 rows");
    const int64_t b128 = ((rows + 127) / 128) * ((n + 127) / 128);
    // batched B needs every block to stay inside one batch slice: tiles never straddle slices when BM divides m,
    // otherwise fall back to one launch per slice (handled by the caller passing rows == m)
#define IGEMM_LAUNCH(BM_, BN_, WM_, WN_, BKB_, OCC_)                                                                         \
    do {                                                                                                                  \
        constexpr size_t lds = (size_t)2 * ((BM_) + (BN_)) * ((BKB_) + 16);                                               \
        auto kern = igemm_kernel<BM_, BN_, WM_, WN_, BKB_, OCC_>;                                                          \
        if (lds > 64 * 1024) LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));               \
        dim3 grid((n + (BN_)-1) / (BN_), (unsigned)((rows + (BM_)-1) / (BM_)));                                           \
        hipLaunchKernelGGL(kern, grid, dim3((WM_) * (WN_) * 64), lds, ctx->stream, aq, wt, rows, n, kp, b_stride,          \
                           m_per_batch, epi);                                                                             \
    } while (0)
    const int64_t b64 = ((rows + 63) / 64) * ((n + 63) / 64);
#ifdef LELE_HIP_LAB
    const int force = lab_int("LELE_HIP_IGEMM_TILE", 0);  // developer override for tile-shape experiments (tools/qlinear_bench.py)
    if (force == 1) IGEMM_LAUNCH(128, 128, 2, 4, 64, 4);
    else if (force == 2) IGEMM_LAUNCH(128, 128, 2, 4, 128, 1);
    else if (force == 3) IGEMM_LAUNCH(256, 128, 4, 2, 64, 2);
    else if (force == 4) IGEMM_LAUNCH(128, 256, 2, 4, 64, 2);
    else if (force == 5) IGEMM_LAUNCH(64, 64, 2, 2, 64, 1);
    else if (force == 6) IGEMM_LAUNCH(128, 128, 2, 2, 64, 2);
    else if (force == 7) IGEMM_LAUNCH(64, 128, 2, 2, 64, 2);
    else if (force == 8) IGEMM_LAUNCH(256, 128, 4, 2, 128, 2);
    else if (force == 9) IGEMM_LAUNCH(128, 128, 2, 4, 64, 6);
    else if (force == 10) IGEMM_LAUNCH(128, 64, 2, 2, 64, 5);
    else if (force == 11) IGEMM_LAUNCH(64, 128, 2, 2, 64, 5);
    else if (force == 12) IGEMM_LAUNCH(128, 128, 2, 2, 64, 4);
    else if (force == 13) IGEMM_LAUNCH(128, 64, 4, 2, 64, 4);
    else if (force == 14) IGEMM_LAUNCH(128, 64, 4, 2, 128, 4);
    else
#endif
    // compute-bound shapes (igemm_big.h): long K in whole 128-byte steps, at least half a chip of 256 x 256 results, 16-byte rows
    if (b_stride == 0 && kp % 128 == 0 && kp >= 1024 && n % 4 == 0 && lab_int("LELE_HIP_IGEMM_BIG", 1) != 0 &&
        rows * kp < (int64_t(1) << 32) && (int64_t)n * kp < (int64_t(1) << 32) &&
        ((rows + 255) / 256) * (((int64_t)n + 255) / 256) * 2 >= (int64_t)ctx->num_cus && ((((uintptr_t)epi.out) | ((uintptr_t)epi.res1) | ((uintptr_t)epi.res2)) & 15) == 0 &&
        !epi.blockstat) {
        dim3 grid((unsigned)((n + 255) / 256), (unsigned)((rows + 255) / 256));
        if (lab_int("LELE_HIP_IGEMM_BIG", 1) == 2) {   // (lab) four waves of 128 x 128
            auto kern = igemm_big_kernel<2>;
            LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), BG_LDS));
            hipLaunchKernelGGL(kern, grid, dim3(256), BG_LDS, ctx->stream, aq, wt, rows, n, kp, epi);
        } else if (lab_int("LELE_HIP_IGEMM_BIG", 1) == 3) {   // (lab) eight waves of 128 x 64, operands through staging registers
            auto kern = igemm_big_kernel<4>;
            LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), BG_LDS));
            hipLaunchKernelGGL(kern, grid, dim3(512), BG_LDS, ctx->stream, aq, wt, rows, n, kp, epi);
        } else {
            auto kern = igemm_big_kernel<4, true>;
            LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), BG_LDS));
            hipLaunchKernelGGL(kern, grid, dim3(512), BG_LDS, ctx->stream, aq, wt, rows, n, kp, epi);
        }
    } else if (b64 < 2 * (int64_t)ctx->num_cus) {
        // small problem (SenseVoice at M = 504): 32x32 tiles, K split over the four waves, operands straight from L2
        dim3 grid((unsigned)((n + 31) / 32), (unsigned)((rows + 31) / 32));
        if (kp <= 512)
            hipLaunchKernelGGL((gemm::igemm_small_kernel<IgemmEpi, 4>), grid, dim3(256), 0, ctx->stream, aq, wt, rows, n, kp, b_stride,
                               m_per_batch, epi);
        else
            hipLaunchKernelGGL((gemm::igemm_small_kernel<IgemmEpi, 16>), grid, dim3(256), 0, ctx->stream, aq, wt, rows, n, kp, b_stride,
                               m_per_batch, epi);
    } else if (b128 >= 2 * ctx->num_cus) {
        // 128-byte K tiles (74 KB of LDS, 2 workgroups per CU): half as many barriers and twice the bytes in flight per K step.
        // Measured on the configs[3] shard shapes (profiles/r02_qlinear_variants.json): 29.3 / 30.0 / 41.1 us for qkv / ffn1 / ffn2
        // against 32.5 / 32.9 / 48.7 us with 64-byte tiles -- the second, nearly empty round of workgroups costs less than that.
        IGEMM_LAUNCH(128, 128, 2, 4, 128, 1);
    } else if (rows <= 32) {
        IGEMM_LAUNCH(32, 128, 1, 4, 64, 1);
    } else if (((rows + 127) / 128) * ((n + 63) / 64) >= ctx->num_cus) {
        // narrow results over many rows (out projection, second feed-forward layer: N = 512 over a batch of utterances): 128 x 64
        // tiles, eight waves, 128-byte K tiles -- 40.0 us against 45.9 us with 64 x 64 tiles at K = 2048, 16.3 against 16.8 at K = 512
        IGEMM_LAUNCH(128, 64, 4, 2, 128, 4);
    } else {
        IGEMM_LAUNCH(64, 64, 2, 2, 64, 1);
    }
#undef IGEMM_LAUNCH
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

// the two passes of the fused two-layer form over the hidden layer's product (128 x 128 tiles, the big-shape configuration)
int launch_igemm_hidden(LeleCtx* ctx, int em, const int8_t* aq, const int8_t* wt, int64_t rows, int n, int kp, int m, const IgemmEpi& epi) {
    constexpr size_t lds = (size_t)2 * (128 + 128) * (128 + 16);
    const dim3 grid((n + 127) / 128, (unsigned)((rows + 127) / 128));
    if (em == 1) {
        auto kern = igemm_kernel<128, 128, 2, 4, 128, 1, 1>;
        LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, ctx->stream, aq, wt, rows, n, kp, (int64_t)0, m, epi);
    } else {
        auto kern = igemm_kernel<128, 128, 2, 4, 128, 1, 2>;
        LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));
        hipLaunchKernelGGL(kern, grid, dim3(512), lds, ctx->stream, aq, wt, rows, n, kp, (int64_t)0, m, epi);
    }
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- register-stationary route (igemm_rs.h): K padded to exactly 512 or 2048 bytes, enough 32 x 32 tiles to fill the chip
int rs_kp(int64_t k) {
    const int64_t kp = (k + 31) & ~int64_t(31);
    return kp == 512 || kp == 2048 ? (int)kp : 0;
}
// enough 32 x 32 tiles, 16-byte stores, 32-bit byte offsets into the result.  (Round 3 asked for 8 tiles per CU: below that the
// quantising pass + small-problem GEMM were faster.  With the rows quantised inside the kernel the route needs no pass in front, and
// one 30 s utterance -- 16 row tiles x 48-64 column tiles -- gains 7 % per forward on it: 2 tiles per CU.)
bool rs_enabled(LeleCtx* ctx, int64_t rows, int64_t n, int per_cu = 2) {
    if (env_int("LELE_HIP_IGEMM_RS", 1) == 0) return false;  // documented switch: 0 = the tiled kernels everywhere
    if (n % 4 || rows * n >= (int64_t(1) << 30)) return false;
    const int64_t units = ((rows + 31) / 32) * ((n + 31) / 32);
    return units >= (int64_t)lab_int("LELE_HIP_IGEMM_RS_MIN", per_cu * ctx->num_cus);
}
// the stand-alone linear.  Measured on one configs[3] shard (tools/rs_bench.py, whole op incl. the row quantisation): N = 1536
// 24.9 us against 28.7 us tiled, N = 2048 27.2 against 29.5 -- but N = 512 23.0 against 18.7 (few column tiles: the
// weights-in-registers prologue is not amortised), and a stand-alone K = 2048 linear 54.9 against 41.3 (the fused feed-forward
// block calls the K-split kernel itself, on a hidden layer that is already in fragment order)
// the quantising loaders (igemm_rs.h, FQ): whole 512-element rows read as float4s.  LELE_HIP_IGEMM_RS_FQ=0 (lab build) keeps the
// separate qrows_frag_kernel pass for A/B timing; both give the same bits.
bool rs_fq(int64_t k, const void* dx) { return k == 512 && (((uintptr_t)dx) & 15) == 0 && lab_int("LELE_HIP_IGEMM_RS_FQ", 1) != 0; }
bool rs_fits(LeleCtx* ctx, int64_t rows, int64_t n, int kp) { return kp == 512 && n >= lab_int("LELE_HIP_IGEMM_RS_MIN_N", 1024) && rs_enabled(ctx, rows, n); }
// the register-stationary kernels read residuals and write results 16 bytes at a time: a device tensor that does not start on a
// 16-byte boundary (a view into a larger buffer) takes the tiled route instead (host tensors are staged into the aligned arena)
bool rs_aligned(const LeleTensor* t) { return !t || t->mem != LELE_MEM_DEVICE || (((uintptr_t)t->data) & 15) == 0; }
// weights in fragment order + column sums: cached for declared-immutable weights, packed into the arena per call otherwise
int frag_weights_of(LeleCtx* ctx, const LeleTensor* w, int k, int n, int kp, FragW* out) {
    const int ks = kp / 32, nt = (n + 31) / 32;
    if (w->mem == LELE_MEM_WEIGHT) return get_frag_weights(ctx, w, k, n, ks, nt, out);
    const void* dw = nullptr;
    LELE_TRY(ctx->dev_ptr(w, &dw));
    void *dwf = nullptr, *dcs = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)nt * ks * 1024, &dwf));
    LELE_TRY(ctx->arena_alloc((size_t)n * 4, &dcs));
    const int64_t threads = (int64_t)nt * 32 * ks * 2;
    hipLaunchKernelGGL(wpack_frag_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, (const float*)dw, k, n, ks, nt,
                       (int8_t*)dwf);
    LELE_TRY(launch_wcolsum(ctx, (const float*)dw, k, n, (int*)dcs));
    LELE_HIP_CHECK(hipGetLastError());
    out->wf = (int8_t*)dwf;
    out->col_sums = (int*)dcs;
    return 0;
}
// f32 rows -> fragment-major i8 [ceil(rows / 32)][kp / 32][1024] in the arena (+ row sums unless rs == NULL)
int launch_qrows_frag(LeleCtx* ctx, const float* dx, int64_t rows, int k, int kp, int m, QParams* prm, int8_t* af, int* rs,
                      const float* partial, int nblk, unsigned* zero_slice) {
    const int64_t nrt = (rows + 31) / 32;
    if (kp == 512)  // a wave per two rows: 4 tiles' worth of workgroups per tile
        hipLaunchKernelGGL((qrows_frag_kernel<32, 1>), dim3((unsigned)(4 * nrt)), dim3(256), 0, ctx->stream, dx, (unsigned)rows, k, m, prm, af, rs,
                           partial, nblk, zero_slice);
    else
        hipLaunchKernelGGL((qrows_frag_kernel<128, 4>), dim3((unsigned)(4 * nrt)), dim3(256), 0, ctx->stream, dx, (unsigned)rows, k, m, prm, af, rs,
                           partial, nblk, zero_slice);
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}
// the grid of igemm_rs_kernel: column blocks of 8 tiles x row ranges, one 640 / 768-thread workgroup per CU, all resident at once
void rs_grid(LeleCtx* ctx, int64_t rows, int64_t n, int* ncb, int* nrr) {
    const int nrt = (int)((rows + 31) / 32), nct = (int)((n + 31) / 32);
    *ncb = (nct + 7) / 8;
    *nrr = std::max(1, std::min(nrt, ctx->num_cus / *ncb));
}
// the most slices of m rows a workgroup's row range can touch (FQ: the loaders keep their parameters in a table of RS_MAXSL entries;
// the fused feed-forward block's range pass keeps four maxima a workgroup)
int64_t rs_max_slices(LeleCtx* ctx, int64_t rows, int64_t n, int64_t m) {
    int ncb, nrr;
    rs_grid(ctx, rows, n, &ncb, &nrr);
    const int64_t nrt = (rows + 31) / 32, tiles = (nrt + nrr - 1) / nrr;
    return rows == m ? 1 : (tiles * 32 + m - 2) / m + 1;
}
struct RsFq {  // what the quantising loaders need besides the epilogue's arguments
    const float* x = nullptr;
    const float* partial = nullptr;
    int nblk = 0;
    unsigned* hpart = nullptr;
    unsigned* sync = nullptr;  // EM 3: the row ranges' arrival counters (rs_sync_counters)
};
// EM 3 (igemm_rs.h: both passes of the feed-forward block's first product in one launch, workgroups waiting for their neighbours):
// only where every workgroup of the launch is resident and nothing else of this library can be waiting beside it -- lane 0 of the ONLY
// live context of the device, no side lane in flight, a grid of at most one workgroup per CU, at most RS_NS_BOTH tiles a workgroup --
// and only with counters that exist already or can be made now (not while capturing).  LELE_HIP_FFN_ONE_LAUNCH=0 keeps the two launches.
unsigned* rs_sync_counters(LeleCtx* ctx, int64_t rows, int64_t n) {
    if (env_int("LELE_HIP_FFN_ONE_LAUNCH", 1) == 0) return nullptr;
    int ncb = 0, nrr = 0;
    rs_grid(ctx, rows, n, &ncb, &nrr);
    const int64_t nrt = (rows + 31) / 32, tiles = (nrt + nrr - 1) / nrr;
    if (tiles > RS_NS_BOTH || nrr > 64 || ncb * nrr > ctx->num_cus || n % 256 != 0) return nullptr;
    if (ctx->lane != 0 || ctx->side_lanes || lele::live_contexts(ctx->device) != 1) return nullptr;
    auto it = ctx->rs_sync.find({ncb, nrr});
    if (it != ctx->rs_sync.end()) return it->second;
    if (ctx->capturing) return nullptr;
    void* p = nullptr;
    // 64 counters, then one 8-byte word {launch tag, maximum} per workgroup and slice (4 a workgroup): cleared once, never again
    const size_t bytes = 256 + (size_t)ncb * nrr * 32;
    // cleared ON THE CONTEXT'S STREAM: the stream does not synchronise with the null stream, and the first launch follows at once
    if (hipMalloc(&p, bytes) != hipSuccess || hipMemsetAsync(p, 0, bytes, ctx->stream) != hipSuccess) {
        (void)hipGetLastError();
        if (p) (void)hipFree(p);
        return nullptr;
    }
    ctx->rs_sync[{ncb, nrr}] = (unsigned*)p;
    return (unsigned*)p;
}
int launch_rs(LeleCtx* ctx, int em, const int8_t* af, const int8_t* wf, int64_t rows, int n, int8_t* hid, const IgemmEpi& epi,
              const RsFq& fq = RsFq()) {
    const float* x_f32 = fq.x;
    RsArgs g{af, wf, (unsigned)rows, n, (int)((rows + 31) / 32), (n + 31) / 32, 0, 0, hid, x_f32, fq.partial, fq.nblk, fq.hpart};
    g.sync = fq.sync;
    g.deverr = ctx->deverr_dev;
#ifdef LELE_HIP_LAB
    g.dbg = nullptr;
    g.ablate = lab_int("LELE_HIP_RS_ABLATE", 0);
    if (const char* e = lab_env("LELE_HIP_RS_STAMPS"))  // a device address (tools/rs_stamps.py), optionally for one mode only
        if (lab_int("LELE_HIP_RS_STAMPS_EM", em) == em) g.dbg = (long long*)(uintptr_t)strtoull(e, nullptr, 0);
#endif
    rs_grid(ctx, rows, n, &g.ncb, &g.nrr);
    const dim3 grid((unsigned)(g.ncb * g.nrr));
    const int nres = epi.res1 ? (epi.res2 ? 2 : 1) : 0;
#define LELE_RS(EM_, NRES_, RELU_)                                                                   \
    do {                                                                                             \
        if (x_f32) {                                                                                 \
            auto kern = igemm_rs_kernel<EM_, NRES_, RELU_, true>;                                     \
            LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), RS_LDS));              \
            hipLaunchKernelGGL(kern, grid, dim3(768), RS_LDS, ctx->stream, g, epi);                   \
        } else {                                                                                     \
            auto kern = igemm_rs_kernel<EM_, NRES_, RELU_>;                                           \
            LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), RS_LDS));              \
            hipLaunchKernelGGL(kern, grid, dim3(640), RS_LDS, ctx->stream, g, epi);                   \
        }                                                                                            \
    } while (0)
    if (em == 3) {
        LELE_REQUIRE(x_f32 && fq.sync && fq.hpart, "launch_rs: the one-launch form needs the quantising loaders and its counters");
        auto kern = igemm_rs_kernel<3, 0, true, true>;
        LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), RS_LDS_BOTH));
        hipLaunchKernelGGL(kern, grid, dim3(768), RS_LDS_BOTH, ctx->stream, g, epi);
    } else if (em == 1) LELE_RS(1, 0, true);
    else if (em == 2) LELE_RS(2, 0, true);
    else if (epi.relu) {
        if (nres == 0) LELE_RS(0, 0, true);
        else if (nres == 1) LELE_RS(0, 1, true);
        else LELE_RS(0, 2, true);
    } else {
        if (nres == 0) LELE_RS(0, 0, false);
        else if (nres == 1) LELE_RS(0, 1, false);
        else LELE_RS(0, 2, false);
    }
#undef LELE_RS
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}
// K = 512, N <= 512: one workgroup a row tile, all columns (igemm_as_kernel)
bool as_fits(LeleCtx* ctx, int64_t rows, int64_t n, int64_t k, const void* dx) {
    if (env_int("LELE_HIP_IGEMM_RS", 1) == 0 || lab_int("LELE_HIP_IGEMM_AS", 1) == 0) return false;
    return k == 512 && n <= 512 && n % 4 == 0 && n >= 4 && rows * n < (int64_t(1) << 30) && (((uintptr_t)dx) & 15) == 0 &&
           rows >= (int64_t)lab_int("LELE_HIP_IGEMM_AS_MIN_ROWS", 256);
}
// a batch: one workgroup a row tile, all (up to 16) column tiles, two a wave; few row tiles (one utterance: 16): 8 or 4 column
// tiles a workgroup, one a wave, so that the grid still covers a good part of the chip
bool as_two(LeleCtx* ctx, int64_t rows, int n) {
    const int64_t nrt = (rows + 31) / 32;
    return nrt * 3 >= (int64_t)ctx->num_cus || (n + 31) / 32 <= 4 || lab_int("LELE_HIP_IGEMM_AS_TPW", 0) == 16;
}
struct AsLn {  // the LayerNorm behind the projection, in the same launch (n == 512, whole rows in a workgroup)
    const float* g = nullptr;
    const float* b = nullptr;
    float eps = 0.0f;
    float* out = nullptr;
    float* rowstat = nullptr;
    // + the FSMN memory block as res1, computed in the kernel (k = 11)
    const float* fs_x = nullptr;
    int fs_pitch = 0;
    const float* fs_w = nullptr;
    const float* fs_b = nullptr;
    int fs_pl = 0;
};
int launch_as(LeleCtx* ctx, const float* x, const int8_t* wf, int64_t rows, int n, const float* partial, int nblk, QParams* prm,
              const IgemmEpi& epi, const AsLn* ln = nullptr) {
    AsArgs g{x, wf, (unsigned)rows, n, (n + 31) / 32, partial, nblk, prm, 16};
    const int64_t nrt = (rows + 31) / 32;
    const bool two = as_two(ctx, rows, n);
    if (!two) g.tpw = lab_int("LELE_HIP_IGEMM_AS_TPW", nrt * 6 >= (int64_t)ctx->num_cus ? 8 : 4);
    const dim3 grid((unsigned)nrt, two ? 1u : (unsigned)((g.nct + g.tpw - 1) / g.tpw));
    const int nres = epi.res1 ? (epi.res2 ? 2 : 1) : 0;
    if (ln) {
        LELE_REQUIRE(two && n == 512 && !epi.relu, "launch_as: the LayerNorm epilogue needs whole 512-wide rows in a workgroup");
        g.ln_g = ln->g, g.ln_b = ln->b, g.ln_eps = ln->eps, g.ln_out = ln->out, g.ln_rowstat = ln->rowstat;
#define LELE_AS_LN(NRES_)                                                                              \
    do {                                                                                               \
        auto kern = igemm_as_kernel<NRES_, false, true, true>;                                          \
        LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), AS_LN_LDS));                 \
        hipLaunchKernelGGL(kern, grid, dim3(512), AS_LN_LDS, ctx->stream, g, epi);                      \
    } while (0)
        if (ln->fs_x) {   // res1 is the memory block of ln->fs_x (epi.res1 is not read), epi.res2 the optional second residual
            g.fs_x = ln->fs_x, g.fs_pitch = ln->fs_pitch, g.fs_w = ln->fs_w, g.fs_b = ln->fs_b;
#define LELE_AS_FS(NRES_)                                                                              \
    do {                                                                                               \
        auto kern = igemm_as_kernel<NRES_, false, true, true, 11>;                                      \
        LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), as_fs_lds(11)));             \
        hipLaunchKernelGGL(kern, grid, dim3(512), as_fs_lds(11), ctx->stream, g, epi);                  \
    } while (0)
            if (epi.res2) LELE_AS_FS(2);
            else LELE_AS_FS(1);
#undef LELE_AS_FS
            LELE_HIP_CHECK(hipGetLastError());
            return 0;
        }
        if (nres == 0) LELE_AS_LN(0);
        else if (nres == 1) LELE_AS_LN(1);
        else LELE_AS_LN(2);
#undef LELE_AS_LN
        LELE_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define LELE_AS(NRES_, RELU_)                                                                                     \
    do {                                                                                                          \
        if (two) hipLaunchKernelGGL((igemm_as_kernel<NRES_, RELU_, true>), grid, dim3(512), 0, ctx->stream, g, epi);   \
        else hipLaunchKernelGGL((igemm_as_kernel<NRES_, RELU_, false>), grid, dim3(512), 0, ctx->stream, g, epi);     \
    } while (0)
    if (epi.relu) {
        if (nres == 0) LELE_AS(0, true);
        else if (nres == 1) LELE_AS(1, true);
        else LELE_AS(2, true);
    } else {
        if (nres == 0) LELE_AS(0, false);
        else if (nres == 1) LELE_AS(1, false);
        else LELE_AS(2, false);
    }
#undef LELE_AS
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}
int launch_rs_ks4(LeleCtx* ctx, const int8_t* af, const int8_t* wf, int64_t rows, int n, const IgemmEpi& epi) {
    RsArgs g{af, wf, (unsigned)rows, n, (int)((rows + 31) / 32), (n + 31) / 32, 0, 0, nullptr};
#ifdef LELE_HIP_LAB
    g.dbg = nullptr;
    g.ablate = 0;
#endif
    g.nrr = std::max(1, std::min(g.nrt, 2 * ctx->num_cus / g.nct));  // two 256-thread workgroups per CU
    const dim3 grid((unsigned)(g.nct * g.nrr));
    const int nres = epi.res1 ? (epi.res2 ? 2 : 1) : 0;
    if (nres == 0) hipLaunchKernelGGL(igemm_rs_ks4_kernel<0>, grid, dim3(256), 0, ctx->stream, g, epi);
    else if (nres == 1) hipLaunchKernelGGL(igemm_rs_ks4_kernel<1>, grid, dim3(256), 0, ctx->stream, g, epi);
    else hipLaunchKernelGGL(igemm_rs_ks4_kernel<2>, grid, dim3(256), 0, ctx->stream, g, epi);
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

// K = 2048, N = 512 over a batch: one workgroup a row tile, all columns, weights streamed (igemm_ask_kernel), optionally + LayerNorm
bool ask_fits(LeleCtx* ctx, int64_t rows, int64_t n, int64_t k) {
    return env_int("LELE_HIP_IGEMM_RS", 1) != 0 && env_int("LELE_HIP_IGEMM_ASK", 1) != 0 && k == 2048 && n == 512 && rows * n < (int64_t(1) << 30) &&
           as_two(ctx, rows, (int)n);
}
int launch_ask(LeleCtx* ctx, const int8_t* af, const int8_t* wf, int64_t rows, const IgemmEpi& epi, const AsLn* ln) {
    AskArgs g{af, wf, (unsigned)rows};
    if (ln) g.ln_g = ln->g, g.ln_b = ln->b, g.ln_eps = ln->eps, g.ln_out = ln->out, g.ln_rowstat = ln->rowstat;
    const dim3 grid((unsigned)((rows + 31) / 32));
    const int nres = epi.res1 ? (epi.res2 ? 2 : 1) : 0;
#define LELE_ASK(NRES_, LN_)                                                                            \
    do {                                                                                                \
        auto kern = igemm_ask_kernel<NRES_, LN_>;                                                        \
        LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(kern), (LN_) ? ASK_LDS : ASK_TILE)); \
        hipLaunchKernelGGL(kern, grid, dim3(512), (LN_) ? ASK_LDS : ASK_TILE, ctx->stream, g, epi);      \
    } while (0)
    if (ln) {
        if (nres == 0) LELE_ASK(0, true);
        else if (nres == 1) LELE_ASK(1, true);
        else LELE_ASK(2, true);
    } else {
        if (nres == 0) LELE_ASK(0, false);
        else if (nres == 1) LELE_ASK(1, false);
        else LELE_ASK(2, false);
    }
#undef LELE_ASK
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

namespace lele {
// {scale, zp, 1/scale, (int)zp} of the joint range of several device arrays (DynamicQuantizeLinear's rule,
// avx/quantization.rs:134-140 == conv2d.rs:2329-2333); prm_dev receives 16 bytes laid out as QParamsDev (common.h)
int quant_params_of(LeleCtx* ctx, const float* const* srcs, const int64_t* lens, int nsrc, void* prm_dev) {
    int64_t total_blocks = 0;
    std::vector<int> nb(nsrc);
    for (int i = 0; i < nsrc; ++i) {
        nb[i] = (int)std::max<int64_t>(1, std::min<int64_t>(256, (lens[i] + 4095) / 4096));
        total_blocks += nb[i];
    }
    void* partial = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)total_blocks * 8, &partial));
    int64_t off = 0;
    for (int i = 0; i < nsrc; ++i) {
        if (lens[i] > 0)
            hipLaunchKernelGGL(qminmax_kernel, dim3(nb[i], 1), dim3(256), 0, ctx->stream, srcs[i], lens[i],
                               (float*)partial + 2 * off);
        else
            nb[i] = 0;
        off += nb[i];
    }
    hipLaunchKernelGGL(qparams_kernel, dim3(1), dim3(64), 0, ctx->stream, (const float*)partial, (int)off, (QParams*)prm_dev,
                       (float*)nullptr, (float*)nullptr);
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}
}  // namespace lele

namespace {
// the producer's {min, max} pairs, if it left any next to this tensor (LayerNorm: one pair per row; the one-launch kernels: a fixed
// number per slice; a small-problem GEMM: one per workgroup, single slice only): slice s then owns `nblk` consecutive pairs and the
// separate range pass over the activation is skipped
void find_partials(LeleCtx* ctx, const LeleTensor* input, int64_t batch, int64_t m, int64_t k, const float** partial, int* nblk) {
    const int64_t rows = batch * m;
    *partial = nullptr;
    *nblk = 0;
    if (input->mem != LELE_MEM_DEVICE || m > 2048) return;
    auto it = ctx->buf_of_data.find(input->data);
    if (it == ctx->buf_of_data.end() || !it->second->rowstat_valid) return;
    const LeleBuf* src = it->second;
    if (src->rowstat_kind == 0 && src->rowstat_len == k && src->rowstat_rows == rows) {
        *partial = src->rowstat;  // one pair per row
        *nblk = (int)m;
    } else if (src->rowstat_kind == 2 && src->rowstat_len == k && src->rowstat_m == m && src->rowstat_batch == batch && src->rowstat_rows % batch == 0) {
        *partial = src->rowstat;  // a fixed number of pairs per slice of m rows (igemm_rs_kernel, the attention kernels)
        *nblk = (int)(src->rowstat_rows / batch);
    } else if (src->rowstat_kind == 1 && batch == 1 && src->rowstat_len == rows * k && src->rowstat_rows <= 4096) {
        *partial = src->rowstat;  // per-workgroup pairs of the GEMM that produced the tensor: one slice, all pairs
        *nblk = (int)src->rowstat_rows;
    }
}

// weight_zero.data.first() as i32 (quantization.rs:100); fetched on the host: it is a scalar attribute
int weight_zero_of(LeleCtx* ctx, const LeleTensor* weight_zero, float* wz) {
    *wz = 0.0f;
    if (weight_zero && numel(weight_zero) > 0) {
        if (weight_zero->mem == LELE_MEM_DEVICE) {
            LELE_HIP_CHECK(hipMemcpyAsync(wz, weight_zero->data, 4, hipMemcpyDeviceToHost, ctx->stream));
            LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
            LELE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        } else {
            *wz = *(const float*)weight_zero->data;
        }
    }
    return 0;
}

// weights as i8 [N][Kp] + column sums (cached for declared-immutable weights)
int packed_weights_of(LeleCtx* ctx, const LeleTensor* weight_int8, int k, int n, int kp, PackedW* pw) {
    const bool cacheable = weight_int8->mem == LELE_MEM_WEIGHT;  // only declared-immutable weights are cached
    auto key_w = std::make_tuple((const void*)weight_int8->data, (size_t)numel(weight_int8) * 4, 201);
    if (cacheable && ctx->weights.count(key_w)) return get_packed_weights(ctx, weight_int8, nullptr, 1, k, n, kp, pw, true);
    const void* dw = nullptr;
    LELE_TRY(ctx->dev_ptr(weight_int8, &dw));
    return get_packed_weights(ctx, weight_int8, (const float*)dw, 1, k, n, kp, pw, cacheable);
}
}  // namespace

extern "C" {

// the LayerNorm epilogue of igemm_as_kernel: whole 512-wide rows in a workgroup, no ReLU, f32 scale / bias of at least 512 values
static bool as_ln_ok(LeleCtx* ctx, int64_t rows, int64_t n, int apply_relu, const LeleTensor* g, const LeleTensor* b) {
    return env_int("LELE_HIP_LN_FUSED", 1) != 0 && n == 512 && !apply_relu && as_two(ctx, rows, (int)n) && g->dtype == LELE_F32 &&
           b->dtype == LELE_F32 && numel(g) >= 512 && numel(b) >= 512;
}
// a LayerNorm over the last axis that the caller wants behind the linear: done in the GEMM's launch where a workgroup holds whole
// rows of the result (`done` says whether it was; the caller runs lele_hip_layer_norm otherwise)
struct LnReq {
    const LeleTensor* g;
    const LeleTensor* b;
    float eps;
    LeleBuf* out;
    bool done = false;
    // optionally the FSMN memory block as the first residual (lele_hip_sanm_out_block): only honoured together with the LayerNorm,
    // `done` then covers both
    const LeleTensor* fs_x = nullptr;   // [B, T, P]
    const LeleTensor* fs_w = nullptr;   // [512, 1, 11]
    const LeleTensor* fs_b = nullptr;
    int64_t fs_pl = 0, fs_off = 0;
};
static int fql_impl(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8, const LeleTensor* weight_scale,
                    const LeleTensor* weight_zero, const LeleTensor* bias, int apply_relu, const LeleTensor* res1,
                    const LeleTensor* res2, LeleBuf* out, int64_t* out_shape, int32_t* out_rank, LnReq* ln = nullptr) {
    LELE_REQUIRE(ctx && input && weight_int8 && weight_scale && out, "fused_quantized_linear: NULL argument");
    LELE_REQUIRE(input->rank >= 2 && weight_int8->rank >= 2, "fused_quantized_linear: rank >= 2 required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t m = input->shape[input->rank - 2], k = input->shape[input->rank - 1];
    const int64_t kw = weight_int8->shape[weight_int8->rank - 2], n = weight_int8->shape[weight_int8->rank - 1];
    LELE_REQUIRE(k == kw, "fused_quantized_linear: K mismatch (%lld vs %lld)", (long long)k, (long long)kw);
    int64_t batch = 1;
    for (int i = 0; i + 2 < input->rank; ++i) batch *= input->shape[i];
    const int64_t ws_len = numel(weight_scale);
    LELE_REQUIRE(ws_len >= 1, "fused_quantized_linear: empty weight_scale");
    LELE_REQUIRE(ws_len <= 1 || ws_len >= n, "fused_quantized_linear: weight_scale has %lld entries for N=%lld",
                 (long long)ws_len, (long long)n);
    const int64_t blen = bias ? numel(bias) : 0;
    LELE_REQUIRE(blen == 0 || blen >= n, "fused_quantized_linear: bias has %lld entries for N=%lld", (long long)blen,
                 (long long)n);
    std::vector<int64_t> shp(input->shape, input->shape + input->rank - 1);
    shp.push_back(n);
    const int64_t rows = batch * m;
    LELE_TRY(out->reserve((size_t)rows * n * 4));
    if (rows == 0 || n == 0) return set_shape_v(out_shape, out_rank, shp);
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dws = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(input, &dx));
    LELE_TRY(ctx->dev_ptr(weight_scale, &dws));
    if (blen) LELE_TRY(ctx->dev_ptr(bias, &db));
    const void *dr1 = nullptr, *dr2 = nullptr;
    if (res1) LELE_TRY(ctx->dev_ptr(res1, &dr1));
    if (res2) LELE_TRY(ctx->dev_ptr(res2, &dr2));
    float wz = 0.0f;
    LELE_TRY(weight_zero_of(ctx, weight_zero, &wz));
    const float* partial = nullptr;
    int nblk = 0;
    find_partials(ctx, input, batch, m, k, &partial, &nblk);
    void* prm = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)batch * sizeof(QParams), &prm));
    LELE_TRY(qprof_mark(ctx, 0));
    if (!partial) LELE_TRY(launch_range(ctx, (const float*)dx, batch, m * k, (QParams*)prm, nullptr, nullptr, &partial, &nblk));
    LELE_TRY(qprof_mark(ctx, 1));

    LELE_REQUIRE(!(ln && ln->fs_x) || (as_fits(ctx, rows, n, k, dx) && rs_aligned(res2) && (((uintptr_t)out->data) & 15) == 0),
                 "internal: the memory block was requested on a route that cannot compute it");
    // ---- register-stationary route (igemm_rs.h): rows -> i8 in fragment order, then the barrier-free GEMM
    if (const int kprs = rs_kp(k); rs_fits(ctx, rows, n, kprs) && rs_aligned(res1) && rs_aligned(res2) && (((uintptr_t)out->data) & 15) == 0) {
        FragW fw;
        LELE_TRY(frag_weights_of(ctx, weight_int8, (int)k, (int)n, kprs, &fw));
        const int64_t nrt = (rows + 31) / 32;
        void *af = nullptr, *rs = nullptr;
        // K = 512 exactly, 16-byte aligned rows: the GEMM's loader waves quantise the f32 rows themselves (igemm_rs.h, FQ) with
        // parameters they reduce from the {min, max} partials at the start of the kernel: nothing is launched in front of it
        const bool fq = rs_fq(k, dx) && rs_max_slices(ctx, rows, n, m) <= RS_MAXSL;
        if (!fq) {
            LELE_TRY(ctx->arena_alloc((size_t)nrt * kprs * 32, &af));
            if (kprs == 512) LELE_TRY(ctx->arena_alloc((size_t)rows * 4, &rs));  // K = 2048: the GEMM sums the rows it loads anyway
            LELE_TRY(launch_qrows_frag(ctx, (const float*)dx, rows, (int)k, kprs, (int)m, (QParams*)prm, (int8_t*)af, (int*)rs, partial, nblk, nullptr));
        }
        LELE_TRY(qprof_mark(ctx, 2));
        IgemmEpi epi{(float*)out->data, rows, n, (int)m, (int)k, (const int*)rs, fw.col_sums, (const QParams*)prm, 0,
                     (int)wz, (const float*)dws, (int)ws_len, blen ? (const float*)db : nullptr, apply_relu, (const float*)dr1,
                     (const float*)dr2};
        const int64_t nstat = nrt * ((n + 31) / 32);
        if (kprs == 512 && batch == 1 && nstat <= 4096) {  // {min, max} per tile for a single-slice consumer (LeleBuf::rowstat kind 1)
            LELE_TRY(out->reserve_rowstat(nstat));
            if ((size_t)nstat <= out->rowstat_cap) epi.blockstat = out->rowstat;
        }
        RsFq fqa;
        if (fq) fqa.x = (const float*)dx, fqa.partial = partial, fqa.nblk = nblk;
        if (kprs == 512) LELE_TRY(launch_rs(ctx, 0, (const int8_t*)af, fw.wf, rows, (int)n, nullptr, epi, fqa));
        else LELE_TRY(launch_rs_ks4(ctx, (const int8_t*)af, fw.wf, rows, (int)n, epi));
        LELE_TRY(qprof_mark(ctx, 3));
        if (epi.blockstat) {
            out->rowstat_rows = nstat;
            out->rowstat_len = rows * n;
            out->rowstat_kind = 1;
            out->rowstat_valid = true;
        }
        return set_shape_v(out_shape, out_rank, shp);
    }

    // ---- activations stationary (igemm_rs.h): few column tiles, K = 512 -- quantise and multiply in one kernel
    if (as_fits(ctx, rows, n, k, dx) && rs_aligned(res1) && rs_aligned(res2) && (((uintptr_t)out->data) & 15) == 0) {
        FragW fw;
        LELE_TRY(frag_weights_of(ctx, weight_int8, (int)k, (int)n, 512, &fw));
        LELE_TRY(qprof_mark(ctx, 2));
        IgemmEpi epi{(float*)out->data, rows, n, (int)m, (int)k, nullptr, fw.col_sums, (const QParams*)prm, 0,
                     (int)wz, (const float*)dws, (int)ws_len, blen ? (const float*)db : nullptr, apply_relu, (const float*)dr1,
                     (const float*)dr2};
        const int64_t nstat = ((rows + 31) / 32) * ((n + 31) / 32);
        if (batch == 1 && nstat <= 4096) {
            LELE_TRY(out->reserve_rowstat(nstat));
            if ((size_t)nstat <= out->rowstat_cap) epi.blockstat = out->rowstat;
        }
        AsLn aln;
        const bool fuse_ln = ln && as_ln_ok(ctx, rows, n, apply_relu, ln->g, ln->b) && ln->out != out;
        LELE_REQUIRE(!(ln && ln->fs_x) || fuse_ln, "internal: the memory block was requested on a route that cannot compute it");
        if (fuse_ln && ln->fs_x) {
            const void *fx = nullptr, *fwp = nullptr, *fbp = nullptr;
            LELE_TRY(ctx->dev_ptr(ln->fs_x, &fx));
            LELE_TRY(ctx->dev_ptr(ln->fs_w, &fwp));
            if (ln->fs_b) LELE_TRY(ctx->dev_ptr(ln->fs_b, &fbp));
            aln.fs_x = (const float*)fx + ln->fs_off, aln.fs_pitch = (int)ln->fs_x->shape[2], aln.fs_w = (const float*)fwp;
            aln.fs_b = (const float*)fbp, aln.fs_pl = (int)ln->fs_pl;
        }
        if (fuse_ln) {
            const void *dg = nullptr, *dbeta = nullptr;
            LELE_TRY(ctx->dev_ptr(ln->g, &dg));
            LELE_TRY(ctx->dev_ptr(ln->b, &dbeta));
            LELE_TRY(ln->out->reserve((size_t)rows * n * 4));
            LELE_TRY(ln->out->reserve_rowstat(rows));
            aln.g = (const float*)dg, aln.b = (const float*)dbeta, aln.eps = ln->eps, aln.out = (float*)ln->out->data;
            aln.rowstat = (size_t)rows <= ln->out->rowstat_cap ? ln->out->rowstat : nullptr;
            epi.out = (float*)out->data;  // (a first reserve of ln->out cannot move `out`: they are different buffers)
        }
        LELE_TRY(launch_as(ctx, (const float*)dx, fw.wf, rows, (int)n, partial, nblk, (QParams*)prm, epi, fuse_ln ? &aln : nullptr));
        if (fuse_ln) {
            ln->done = true;
            if (aln.rowstat) {   // as lele_hip_layer_norm leaves them: one {min, max} pair per row
                ln->out->rowstat_rows = rows;
                ln->out->rowstat_len = n;
                ln->out->rowstat_kind = 0;
                ln->out->rowstat_valid = true;
            }
        }
        LELE_TRY(qprof_mark(ctx, 3));
        if (epi.blockstat) {
            out->rowstat_rows = nstat;
            out->rowstat_len = rows * n;
            out->rowstat_kind = 1;
            out->rowstat_valid = true;
        }
        return set_shape_v(out_shape, out_rank, shp);
    }

    // ---- three-kernel chain: rows -> i8 (+ row sums) in HBM, then the tiled i8 GEMM
    const int kp = (int)((k + 15) & ~int64_t(15));
    PackedW pw;
    LELE_TRY(packed_weights_of(ctx, weight_int8, (int)k, (int)n, kp, &pw));
    void *aq = nullptr, *rs = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)rows * kp, &aq));
    LELE_TRY(ctx->arena_alloc((size_t)rows * 4, &rs));
    if (rows <= 2048 || (kp <= 2048 && !lab_int("LELE_HIP_QROWS_STREAM", 0)))  // the whole row in flight at once (8 x 16 bytes per lane)
        hipLaunchKernelGGL((qrows_kernel<0, true>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)dx,
                           rows, (int)k, kp, (int)m, (QParams*)prm, (int8_t*)aq, (int*)rs, partial, nblk, (unsigned*)nullptr, (int*)nullptr);
    else
        hipLaunchKernelGGL((qrows_kernel<0, false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)dx,
                           rows, (int)k, kp, (int)m, (QParams*)prm, (int8_t*)aq, (int*)rs, partial, nblk, (unsigned*)nullptr, (int*)nullptr);
    LELE_TRY(qprof_mark(ctx, 2));
    IgemmEpi epi{(float*)out->data, rows, n, (int)m, (int)k, (const int*)rs, pw.col_sums, (const QParams*)prm, 0,
                 (int)wz, (const float*)dws, (int)ws_len, blen ? (const float*)db : nullptr, apply_relu, (const float*)dr1,
                 (const float*)dr2};
    // small-problem kernel: publish one {min, max} pair per workgroup next to the result (for the linear that may follow)
    const int64_t b64 = ((rows + 63) / 64) * ((n + 63) / 64), nstat = ((rows + 31) / 32) * ((n + 31) / 32);
    const bool stats = b64 < 2 * (int64_t)ctx->num_cus && nstat <= 4096;
    if (stats) {
        LELE_TRY(out->reserve_rowstat(nstat));
        if ((size_t)nstat <= out->rowstat_cap) epi.blockstat = out->rowstat;
    }
    LELE_TRY(launch_igemm(ctx, (const int8_t*)aq, pw.wt, rows, (int)n, kp, 0, (int)m, epi));
    LELE_TRY(qprof_mark(ctx, 3));
    if (epi.blockstat) {
        out->rowstat_rows = nstat;
        out->rowstat_len = rows * n;
        out->rowstat_kind = 1;
        out->rowstat_valid = true;
    }
    return set_shape_v(out_shape, out_rank, shp);
}

int lele_hip_fused_quantized_linear(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8,
                                    const LeleTensor* weight_scale, const LeleTensor* weight_zero,
                                    const LeleTensor* bias, int apply_relu, LeleBuf* out, int64_t* out_shape,
                                    int32_t* out_rank) {
    return fql_impl(ctx, input, weight_int8, weight_scale, weight_zero, bias, apply_relu, nullptr, nullptr, out, out_shape, out_rank);
}

int lele_hip_binary(LeleCtx* ctx, int op, const LeleTensor* a, const LeleTensor* b, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

int lele_hip_fused_quantized_linear_residual(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8,
                                             const LeleTensor* weight_scale, const LeleTensor* weight_zero, const LeleTensor* bias,
                                             int apply_relu, const LeleTensor* res1, const LeleTensor* res2, LeleBuf* out,
                                             int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && input && weight_int8 && res1 && out, "fused_quantized_linear_residual: NULL argument");
    LELE_REQUIRE(res1->dtype == LELE_F32 && (!res2 || res2->dtype == LELE_F32), "fused_quantized_linear_residual: f32 residuals required");
    int64_t on = weight_int8->shape[weight_int8->rank - 1];
    for (int i = 0; i + 1 < input->rank; ++i) on *= input->shape[i];
    // same shape as the linear's result (leading unit dimensions aside): folded into the GEMM's store
    auto same_shape = [&](const LeleTensor* t) {
        if (numel(t) != on) return false;
        const int r_lin = input->rank;
        if (t->rank > r_lin) return false;  // a higher-rank residual changes the result's rank: the generic path
        for (int d = 0; d < t->rank; ++d) {
            const int64_t want = d == 0 ? weight_int8->shape[weight_int8->rank - 1] : input->shape[r_lin - 1 - d];
            if (t->shape[t->rank - 1 - d] != want) return false;
        }
        return true;
    };
    if (same_shape(res1) && (!res2 || same_shape(res2)))
        return fql_impl(ctx, input, weight_int8, weight_scale, weight_zero, bias, apply_relu, res1, res2, out, out_shape, out_rank);
    // broadcasting residuals: the node sequence this op stands for -- the plain linear, then one `add` per residual with full
    // numpy broadcasting (a residual may broadcast OUTWARD: intermediates go to library-owned temporaries, the last sum to `out`)
    int64_t sh[LELE_MAX_RANK];
    int32_t r = 0;
    const int nres = res2 ? 2 : 1;
    LeleBuf *t0 = nullptr, *t1 = nullptr;
    LELE_TRY(ctx->tmp_buf(0, &t0));
    LELE_TRY(ctx->tmp_buf(1, &t1));
    LELE_TRY(fql_impl(ctx, input, weight_int8, weight_scale, weight_zero, bias, apply_relu, nullptr, nullptr, t0, sh, &r));
    LeleBuf* cur = t0;
    for (int i = 0; i < nres; ++i) {
        const LeleTensor* res = i == 0 ? res1 : res2;
        LeleBuf* dst = i == nres - 1 ? out : (cur == t0 ? t1 : t0);
        LeleTensor t{cur->data, sh, r, LELE_F32, LELE_MEM_DEVICE};
        int64_t sh2[LELE_MAX_RANK];
        int32_t r2 = 0;
        LELE_TRY(lele_hip_binary(ctx, 0 /* add */, &t, res, dst, sh2, &r2));
        for (int d = 0; d < r2; ++d) sh[d] = sh2[d];
        r = r2;
        cur = dst;
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(sh, sh + r));
}

int lele_hip_layer_norm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, const LeleTensor* bias, int32_t axis, float epsilon,
                        LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* x1 = fused_quantized_linear[_residual](input, W.., relu, res1, res2);  x1n = layer_norm(x1, ln_scale, ln_bias, axis = -1, epsilon)
 * -- a projection, the Adds behind it and the LayerNorm that reads their sum (norm.rs:226 -> avx/norm.rs:10-133), bit for bit the
 * two calls.  Where a workgroup of the GEMM holds whole rows of the result (K = 512, N = 512 over a batch: igemm_as_kernel) the
 * normalisation runs in its epilogue; everywhere else the two calls are issued here.  res1 may be NULL (then res2 must be). */
int lele_hip_fused_quantized_linear_residual_ln(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8,
                                                const LeleTensor* weight_scale, const LeleTensor* weight_zero, const LeleTensor* bias,
                                                int apply_relu, const LeleTensor* res1, const LeleTensor* res2, const LeleTensor* ln_scale,
                                                const LeleTensor* ln_bias, float epsilon, LeleBuf* out, LeleBuf* ln_out, int64_t* out_shape,
                                                int32_t* out_rank) {
    LELE_REQUIRE(ctx && input && weight_int8 && ln_scale && ln_bias && out && ln_out, "fused_quantized_linear_residual_ln: NULL argument");
    LELE_REQUIRE(out != ln_out, "fused_quantized_linear_residual_ln: the sum and its normalised form need two buffers");
    LELE_REQUIRE(res1 || !res2, "fused_quantized_linear_residual_ln: res2 without res1");
    LELE_REQUIRE(input->rank >= 2 && weight_int8->rank >= 2, "fused_quantized_linear: rank >= 2 required");
    LnReq ln{ln_scale, ln_bias, epsilon, ln_out};
    int64_t sh[LELE_MAX_RANK];
    int32_t r = 0;
    bool direct = true;   // same-shape residuals: the call that may normalise in its epilogue
    if (res1) {
        int64_t on = weight_int8->shape[weight_int8->rank - 1];
        for (int i = 0; i + 1 < input->rank; ++i) on *= input->shape[i];
        auto same_shape = [&](const LeleTensor* t) {
            if (t->dtype != LELE_F32 || numel(t) != on || t->rank > input->rank) return false;
            for (int d = 0; d < t->rank; ++d)
                if (t->shape[t->rank - 1 - d] != (d == 0 ? weight_int8->shape[weight_int8->rank - 1] : input->shape[input->rank - 1 - d])) return false;
            return true;
        };
        direct = same_shape(res1) && (!res2 || same_shape(res2));
    }
    if (direct) LELE_TRY(fql_impl(ctx, input, weight_int8, weight_scale, weight_zero, bias, apply_relu, res1, res2, out, sh, &r, &ln));
    else LELE_TRY(lele_hip_fused_quantized_linear_residual(ctx, input, weight_int8, weight_scale, weight_zero, bias, apply_relu, res1, res2, out, sh, &r));
    if (!ln.done) {
        LeleTensor x1{out->data, sh, r, LELE_F32, LELE_MEM_DEVICE};
        int64_t sh2[LELE_MAX_RANK];
        int32_t r2 = 0;
        LELE_TRY(lele_hip_layer_norm(ctx, &x1, ln_scale, ln_bias, -1, epsilon, ln_out, sh2, &r2));
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(sh, sh + r));
}

int lele_hip_depthwise_conv1d_tlc(LeleCtx* ctx, const LeleTensor* x, int64_t x_offset, const LeleTensor* w, const LeleTensor* bias,
                                  int64_t pad_left, int64_t pad_right, int relu, int add_input, LeleBuf* out, int64_t* out_shape, int32_t* out_rank);

/* The output half of a SAN-M attention block (SenseVoice's encoder layer) as ONE call:
 *   mem = depthwise_conv1d_tlc(v_src, fsmn_w, fsmn_bias, x_offset, pad_left, pad_right, relu = 0, add_input = 1)     (FSMN memory + v)
 *   out = fused_quantized_linear_residual(input, W.., apply_relu, mem, res2);  ln_out = layer_norm(out, ln_scale, ln_bias, -1, epsilon)
 * bit for bit those three calls.  With K = N = 512 over a batch of utterances the GEMM's workgroup (one 32-row tile, all columns)
 * computes the memory block of its rows from a (32 + k - 1)-row window of v in LDS and normalises its rows in the epilogue:
 * three launches become one; every other shape issues the three calls. */
int lele_hip_sanm_out_block(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* weight_int8, const LeleTensor* weight_scale,
                            const LeleTensor* weight_zero, const LeleTensor* bias, int apply_relu, const LeleTensor* v_src,
                            const LeleTensor* fsmn_w, const LeleTensor* fsmn_bias, int64_t x_offset, int64_t pad_left, int64_t pad_right,
                            const LeleTensor* res2, const LeleTensor* ln_scale, const LeleTensor* ln_bias, float epsilon, LeleBuf* out,
                            LeleBuf* ln_out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && input && weight_int8 && v_src && fsmn_w && ln_scale && ln_bias && out && ln_out, "sanm_out_block: NULL argument");
    LELE_REQUIRE(out != ln_out, "sanm_out_block: the sum and its normalised form need two buffers");
    LELE_REQUIRE(input->rank >= 2 && weight_int8->rank >= 2, "fused_quantized_linear: rank >= 2 required");
    const int64_t m = input->shape[input->rank - 2], k = input->shape[input->rank - 1], n = weight_int8->shape[weight_int8->rank - 1];
    int64_t batch = 1;
    for (int i = 0; i + 2 < input->rank; ++i) batch *= input->shape[i];
    const int64_t rows = batch * m;
    // the one-launch form: the projection on igemm_as_kernel with whole rows, v_src [B, T, P] with B * T the projection's rows and T its
    // slice length, 512 channels from a 16-byte aligned offset, k = 11 with the output as long as the input, res2 of the result's shape
    auto aligned16 = [](const LeleTensor* t) { return t->mem != LELE_MEM_DEVICE || (((uintptr_t)t->data) & 15) == 0; };
    bool fused = env_int("LELE_HIP_FSMN_FUSED", 1) != 0 && weight_int8->shape[weight_int8->rank - 2] == k && rows > 0 &&
                 as_fits(ctx, rows, n, k, input->mem == LELE_MEM_DEVICE ? input->data : nullptr) && !rs_fits(ctx, rows, n, rs_kp(k)) &&
                 as_ln_ok(ctx, rows, n, apply_relu, ln_scale, ln_bias) && v_src->rank == 3 && v_src->dtype == LELE_F32 && fsmn_w->rank == 3 &&
                 fsmn_w->dtype == LELE_F32 && fsmn_w->shape[0] == 512 && fsmn_w->shape[1] == 1 && fsmn_w->shape[2] == 11 &&
                 v_src->shape[0] * v_src->shape[1] == rows && v_src->shape[1] == m && m >= 32 && x_offset >= 0 && x_offset % 4 == 0 &&
                 x_offset + 512 <= v_src->shape[2] && v_src->shape[2] % 4 == 0 && aligned16(v_src) && pad_left == 5 && pad_right == 5 && (!fsmn_bias || (fsmn_bias->dtype == LELE_F32 && numel(fsmn_bias) >= 512)) &&
                 v_src->shape[2] * rows < (int64_t(1) << 30);
    if (fused && res2) {
        fused = res2->dtype == LELE_F32 && numel(res2) == rows * n && res2->rank <= input->rank && aligned16(res2);
        for (int d = 0; fused && d < res2->rank; ++d)
            fused = res2->shape[res2->rank - 1 - d] == (d == 0 ? n : input->shape[input->rank - 1 - d]);
    }
    int64_t sh[LELE_MAX_RANK];
    int32_t r = 0;
    if (fused) {
        LELE_TRY(out->reserve((size_t)rows * n * 4));
        fused = (((uintptr_t)out->data) & 15) == 0;
    }
    if (fused) {
        LnReq ln{ln_scale, ln_bias, epsilon, ln_out};
        ln.fs_x = v_src, ln.fs_w = fsmn_w, ln.fs_b = fsmn_bias, ln.fs_pl = pad_left, ln.fs_off = x_offset;
        // res1 of the kernel is the memory block it computes; the caller's res2 stays the second residual.  fql_impl's NRES counts
        // operands: hand it `input` as a stand-in for res1 (never read in this form)
        LELE_TRY(fql_impl(ctx, input, weight_int8, weight_scale, weight_zero, bias, apply_relu, input, res2, out, sh, &r, &ln));
        LELE_REQUIRE(ln.done, "internal: sanm_out_block's one-launch form did not run");
        return set_shape_v(out_shape, out_rank, std::vector<int64_t>(sh, sh + r));
    }
    LeleBuf* mem = nullptr;
    LELE_TRY(ctx->tmp_buf(2, &mem));
    int64_t msh[LELE_MAX_RANK];
    int32_t mr = 0;
    LELE_TRY(lele_hip_depthwise_conv1d_tlc(ctx, v_src, x_offset, fsmn_w, fsmn_bias, pad_left, pad_right, 0, 1, mem, msh, &mr));
    LeleTensor mt{mem->data, msh, mr, LELE_F32, LELE_MEM_DEVICE};
    return lele_hip_fused_quantized_linear_residual_ln(ctx, input, weight_int8, weight_scale, weight_zero, bias, apply_relu, &mt, res2, ln_scale,
                                                       ln_bias, epsilon, out, ln_out, out_shape, out_rank);
}

/* Two quantised linears with a ReLU between them (the feed-forward block of a transformer layer):
 *     out = fused_quantized_linear[_residual](fused_quantized_linear(input, W1.., relu = 1), W2.., relu2, res1, res2)
 * bit for bit.  When the hidden layer is large the f32 hidden tensor is never stored: its product runs twice on the matrix cores
 * (igemm_kernel EM 1: range only; EM 2: quantise with that range, i8 + row sums), then the second GEMM consumes the i8 rows. */
static int ffn_impl(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* w1_int8, const LeleTensor* w1_scale,
                    const LeleTensor* w1_zero, const LeleTensor* b1, const LeleTensor* w2_int8,
                    const LeleTensor* w2_scale, const LeleTensor* w2_zero, const LeleTensor* b2, int apply_relu2,
                    const LeleTensor* res1, const LeleTensor* res2, LeleBuf* out, int64_t* out_shape, int32_t* out_rank, LnReq* ln) {
    LELE_REQUIRE(ctx && input && w1_int8 && w1_scale && w2_int8 && w2_scale && out, "fused_ffn_quantized: NULL argument");
    LELE_REQUIRE(input->rank >= 2 && w1_int8->rank >= 2 && w2_int8->rank >= 2, "fused_ffn_quantized: rank >= 2 required");
    LELE_REQUIRE(!res2 || res1, "fused_ffn_quantized: res2 without res1");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t m = input->shape[input->rank - 2], k1 = input->shape[input->rank - 1];
    const int64_t n1 = w1_int8->shape[w1_int8->rank - 1], k2 = w2_int8->shape[w2_int8->rank - 2], n2 = w2_int8->shape[w2_int8->rank - 1];
    int64_t batch = 1;
    for (int i = 0; i + 2 < input->rank; ++i) batch *= input->shape[i];
    const int64_t rows = batch * m;
    const int64_t ws1_len = numel(w1_scale), b1_len = b1 ? numel(b1) : 0;
    // the fused route: a hidden layer of whole 128-column tiles (= whole SIMD bodies of its quantiser), slices of at least one tile
    // of rows, enough tiles to fill the chip, same-shape residuals; everything else runs the two calls it stands for
    const bool args_ok = w1_int8->shape[w1_int8->rank - 2] == k1 && k2 == n1 && rows < (int64_t(1) << 31) && ws1_len >= 1 &&
                         (ws1_len == 1 || ws1_len >= n1) && (b1_len == 0 || b1_len >= n1) && n2 >= 1;
    // register-stationary route (igemm_rs.h): K = 512 into a hidden layer of exactly 2048 columns, whose i8 form is written in the
    // fragment order of the second product and read once
    // (the second product's threshold is ONE tile per CU: at one 30 s utterance -- 16 x 16 tiles -- the fused block replaces ffn1 with an f32
    // hidden layer | row quantisation | small-problem GEMM: a configs[2] forward 5.33 -> 5.05 ms, round 6)
    const bool rs_route = env_int("LELE_HIP_FFN_FUSED", 1) != 0 && args_ok && rs_kp(k1) == 512 && n1 == 2048 && rs_enabled(ctx, rows, n1) &&
                          rs_enabled(ctx, rows, n2, 1) && rs_aligned(res1) && rs_aligned(res2);
    bool fused = rs_route || (env_int("LELE_HIP_FFN_FUSED", 1) != 0 && args_ok && n1 % 128 == 0 && m >= 128 &&
                              ((rows + 127) / 128) * (n1 / 128) >= 2 * (int64_t)ctx->num_cus);
    if (fused && res1) {
        int64_t on = n2 * rows;
        auto same = [&](const LeleTensor* t) {
            if (t->dtype != LELE_F32 || numel(t) != on || t->rank > input->rank) return false;
            for (int d = 0; d < t->rank; ++d)
                if (t->shape[t->rank - 1 - d] != (d == 0 ? n2 : input->shape[input->rank - 1 - d])) return false;
            return true;
        };
        fused = same(res1) && (!res2 || same(res2));
    }
    if (!fused) {
        LeleBuf* hid = nullptr;
        LELE_TRY(ctx->tmp_buf(2, &hid));
        int64_t sh[LELE_MAX_RANK];
        int32_t r = 0;
        LELE_TRY(fql_impl(ctx, input, w1_int8, w1_scale, w1_zero, b1, 1, nullptr, nullptr, hid, sh, &r));
        LeleTensor h{hid->data, sh, r, LELE_F32, LELE_MEM_DEVICE};
        if (res1) return lele_hip_fused_quantized_linear_residual(ctx, &h, w2_int8, w2_scale, w2_zero, b2, apply_relu2, res1, res2, out, out_shape, out_rank);
        return fql_impl(ctx, &h, w2_int8, w2_scale, w2_zero, b2, apply_relu2, nullptr, nullptr, out, out_shape, out_rank);
    }
    const int64_t ws2_len = numel(w2_scale), b2_len = b2 ? numel(b2) : 0;
    LELE_REQUIRE(ws2_len >= 1 && (ws2_len == 1 || ws2_len >= n2), "fused_ffn_quantized: weight_scale has %lld entries for N=%lld", (long long)ws2_len, (long long)n2);
    LELE_REQUIRE(b2_len == 0 || b2_len >= n2, "fused_ffn_quantized: bias has %lld entries for N=%lld", (long long)b2_len, (long long)n2);
    std::vector<int64_t> shp(input->shape, input->shape + input->rank - 1);
    shp.push_back(n2);
    LELE_TRY(out->reserve((size_t)rows * n2 * 4));
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dws1 = nullptr, *db1 = nullptr, *dws2 = nullptr, *db2 = nullptr, *dr1 = nullptr, *dr2 = nullptr;
    LELE_TRY(ctx->dev_ptr(input, &dx));
    LELE_TRY(ctx->dev_ptr(w1_scale, &dws1));
    LELE_TRY(ctx->dev_ptr(w2_scale, &dws2));
    if (b1_len) LELE_TRY(ctx->dev_ptr(b1, &db1));
    if (b2_len) LELE_TRY(ctx->dev_ptr(b2, &db2));
    if (res1) LELE_TRY(ctx->dev_ptr(res1, &dr1));
    if (res2) LELE_TRY(ctx->dev_ptr(res2, &dr2));
    float wz1 = 0.0f, wz2 = 0.0f;
    LELE_TRY(weight_zero_of(ctx, w1_zero, &wz1));
    LELE_TRY(weight_zero_of(ctx, w2_zero, &wz2));
    const float* partial = nullptr;
    int nblk = 0;
    find_partials(ctx, input, batch, m, k1, &partial, &nblk);
    void *prm1 = nullptr, *prm2 = nullptr, *aq1 = nullptr, *rs1 = nullptr, *aq2 = nullptr, *rs2 = nullptr, *hmax = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)batch * sizeof(QParams), &prm1));
    LELE_TRY(ctx->arena_alloc((size_t)batch * sizeof(QParams), &prm2));
    LELE_TRY(qprof_mark(ctx, 0));
    if (!partial) LELE_TRY(launch_range(ctx, (const float*)dx, batch, m * k1, (QParams*)prm1, nullptr, nullptr, &partial, &nblk));
    LELE_TRY(qprof_mark(ctx, 1));
    if (rs_route) {
        FragW fw1, fw2;
        LELE_TRY(frag_weights_of(ctx, w1_int8, (int)k1, (int)n1, 512, &fw1));
        LELE_TRY(frag_weights_of(ctx, w2_int8, (int)k2, (int)n2, 2048, &fw2));
        const int64_t nrt = (rows + 31) / 32;
        void *af1 = nullptr, *hid = nullptr;
        // both passes of the first product quantise the f32 rows in their loader waves, with parameters reduced inside the kernels; the
        // range pass leaves four maxima a workgroup (plain stores: nothing to clear) which the quantise pass's loaders reduce
        const bool fq = rs_fq(k1, dx) && rs_max_slices(ctx, rows, n1, m) <= 4;
        LELE_TRY(ctx->arena_alloc((size_t)nrt * 2048 * 32, &hid));
        LELE_TRY(ctx->arena_alloc((size_t)batch * 4, &hmax));
        RsFq fqa;
        if (fq) {
            int ncb = 0, nrr = 0;
            rs_grid(ctx, rows, n1, &ncb, &nrr);
            void* hpart = nullptr;
            LELE_TRY(ctx->arena_alloc((size_t)ncb * nrr * 16, &hpart));
            fqa.x = (const float*)dx, fqa.partial = partial, fqa.nblk = nblk, fqa.hpart = (unsigned*)hpart;
            fqa.sync = rs_sync_counters(ctx, rows, n1);
        } else {
            LELE_TRY(ctx->arena_alloc((size_t)nrt * 512 * 32, &af1));
            LELE_TRY(ctx->arena_alloc((size_t)rows * 4, &rs1));
            // rows -> i8 for the first product; the same launch clears the per-slice maxima the range pass adds into
            LELE_TRY(launch_qrows_frag(ctx, (const float*)dx, rows, (int)k1, 512, (int)m, (QParams*)prm1, (int8_t*)af1, (int*)rs1, partial, nblk,
                                       (unsigned*)hmax));
        }
        LELE_TRY(qprof_mark(ctx, 2));
        // (tests) LELE_HIP_FFN_ONE_LAUNCH=2: fail instead of falling back to the two launches, so that a test of the one-launch form tests it
        LELE_REQUIRE(env_int("LELE_HIP_FFN_ONE_LAUNCH", 1) != 2 || fqa.sync,
                     "fused_ffn_quantized: LELE_HIP_FFN_ONE_LAUNCH=2, but this call cannot take the one-launch form (K = 512 exactly, 16-byte aligned "
                     "rows, at most 4 slices and %d row tiles a workgroup, lane 0 of the only context of the device, counters made before a capture)",
                     RS_NS_BOTH);
        IgemmEpi e1{nullptr, rows, n1, (int)m, (int)k1, (const int*)rs1, fw1.col_sums, (const QParams*)prm1, 0, (int)wz1, (const float*)dws1,
                    (int)ws1_len, b1_len ? (const float*)db1 : nullptr, 1};
        e1.slice_max = (unsigned*)hmax;
        if (fqa.sync) {  // both passes in one launch (igemm_rs.h, EM 3)
            e1.q_prm = (QParams*)prm2;
            LELE_TRY(launch_rs(ctx, 3, nullptr, fw1.wf, rows, (int)n1, (int8_t*)hid, e1, fqa));
        } else {
            LELE_TRY(launch_rs(ctx, 1, (const int8_t*)af1, fw1.wf, rows, (int)n1, nullptr, e1, fqa));   // range of the ReLU result per slice
            e1.q_prm = (QParams*)prm2;
            LELE_TRY(launch_rs(ctx, 2, (const int8_t*)af1, fw1.wf, rows, (int)n1, (int8_t*)hid, e1, fqa));  // the result again, as the next operand
        }
        IgemmEpi e2{(float*)out->data, rows, n2, (int)m, (int)k2, nullptr, fw2.col_sums, (const QParams*)prm2, 0, (int)wz2,
                    (const float*)dws2, (int)ws2_len, b2_len ? (const float*)db2 : nullptr, apply_relu2, (const float*)dr1, (const float*)dr2};
        // the second product: one column tile a workgroup over a range of row tiles (igemm_rs_ks4_kernel), or -- where the caller wants
        // the LayerNorm of the result as well -- one row tile a workgroup with all 512 columns and the normalisation in the epilogue
        const bool fuse_ln = ln && ask_fits(ctx, rows, n2, k2) && as_ln_ok(ctx, rows, n2, apply_relu2, ln->g, ln->b) && ln->out != out &&
                             (((uintptr_t)out->data) & 15) == 0;
        if (fuse_ln || (lab_int("LELE_HIP_IGEMM_ASK_ALWAYS", 0) != 0 && ask_fits(ctx, rows, n2, k2) && (((uintptr_t)out->data) & 15) == 0)) {
            AsLn aln;
            if (fuse_ln) {
                const void *dg = nullptr, *dbeta = nullptr;
                LELE_TRY(ctx->dev_ptr(ln->g, &dg));
                LELE_TRY(ctx->dev_ptr(ln->b, &dbeta));
                LELE_TRY(ln->out->reserve((size_t)rows * n2 * 4));
                LELE_TRY(ln->out->reserve_rowstat(rows));
                aln.g = (const float*)dg, aln.b = (const float*)dbeta, aln.eps = ln->eps, aln.out = (float*)ln->out->data;
                aln.rowstat = (size_t)rows <= ln->out->rowstat_cap ? ln->out->rowstat : nullptr;
            }
            LELE_TRY(launch_ask(ctx, (const int8_t*)hid, fw2.wf, rows, e2, fuse_ln ? &aln : nullptr));
            if (fuse_ln) {
                ln->done = true;
                if (aln.rowstat) {
                    ln->out->rowstat_rows = rows;
                    ln->out->rowstat_len = n2;
                    ln->out->rowstat_kind = 0;
                    ln->out->rowstat_valid = true;
                }
            }
        } else {
            LELE_TRY(launch_rs_ks4(ctx, (const int8_t*)hid, fw2.wf, rows, (int)n2, e2));
        }
        LELE_TRY(qprof_mark(ctx, 3));
        return set_shape_v(out_shape, out_rank, shp);
    }
    const int kp1 = (int)((k1 + 15) & ~int64_t(15)), kp2 = (int)n1;
    PackedW pw1, pw2;
    LELE_TRY(packed_weights_of(ctx, w1_int8, (int)k1, (int)n1, kp1, &pw1));
    LELE_TRY(packed_weights_of(ctx, w2_int8, (int)k2, (int)n2, kp2, &pw2));
    LELE_TRY(ctx->arena_alloc((size_t)rows * kp1, &aq1));
    LELE_TRY(ctx->arena_alloc((size_t)rows * 4, &rs1));
    LELE_TRY(ctx->arena_alloc((size_t)rows * kp2, &aq2));
    LELE_TRY(ctx->arena_alloc((size_t)rows * 4, &rs2));
    LELE_TRY(ctx->arena_alloc((size_t)batch * 4, &hmax));
    // rows -> i8 for the first GEMM; the same launch clears the accumulators of the two hidden-layer passes
    if (kp1 <= 2048 && !lab_int("LELE_HIP_QROWS_STREAM", 0))
        hipLaunchKernelGGL((qrows_kernel<0, true>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)dx, rows,
                           (int)k1, kp1, (int)m, (QParams*)prm1, (int8_t*)aq1, (int*)rs1, partial, nblk, (unsigned*)hmax, (int*)rs2);
    else
        hipLaunchKernelGGL((qrows_kernel<0, false>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)dx, rows,
                           (int)k1, kp1, (int)m, (QParams*)prm1, (int8_t*)aq1, (int*)rs1, partial, nblk, (unsigned*)hmax, (int*)rs2);
    LELE_TRY(qprof_mark(ctx, 2));
    IgemmEpi e1{nullptr, rows, n1, (int)m, (int)k1, (const int*)rs1, pw1.col_sums, (const QParams*)prm1, 0, (int)wz1, (const float*)dws1,
                (int)ws1_len, b1_len ? (const float*)db1 : nullptr, 1};
    e1.slice_max = (unsigned*)hmax;
    LELE_TRY(launch_igemm_hidden(ctx, 1, (const int8_t*)aq1, pw1.wt, rows, (int)n1, kp1, (int)m, e1));
    e1.q_out = (int8_t*)aq2;
    e1.q_rowsum = (int*)rs2;
    e1.q_prm = (QParams*)prm2;
    LELE_TRY(launch_igemm_hidden(ctx, 2, (const int8_t*)aq1, pw1.wt, rows, (int)n1, kp1, (int)m, e1));
    IgemmEpi e2{(float*)out->data, rows, n2, (int)m, (int)k2, (const int*)rs2, pw2.col_sums, (const QParams*)prm2, 0, (int)wz2,
                (const float*)dws2, (int)ws2_len, b2_len ? (const float*)db2 : nullptr, apply_relu2, (const float*)dr1, (const float*)dr2};
    LELE_TRY(launch_igemm(ctx, (const int8_t*)aq2, pw2.wt, rows, (int)n2, kp2, 0, (int)m, e2));
    LELE_TRY(qprof_mark(ctx, 3));
    return set_shape_v(out_shape, out_rank, shp);
}

int lele_hip_fused_ffn_quantized(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* w1_int8, const LeleTensor* w1_scale,
                                 const LeleTensor* w1_zero, const LeleTensor* b1, const LeleTensor* w2_int8,
                                 const LeleTensor* w2_scale, const LeleTensor* w2_zero, const LeleTensor* b2, int apply_relu2,
                                 const LeleTensor* res1, const LeleTensor* res2, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    return ffn_impl(ctx, input, w1_int8, w1_scale, w1_zero, b1, w2_int8, w2_scale, w2_zero, b2, apply_relu2, res1, res2, out, out_shape, out_rank, nullptr);
}

/* out = fused_ffn_quantized(...);  ln_out = layer_norm(out, ln_scale, ln_bias, -1, epsilon) -- a transformer layer's feed-forward block,
 * its residual Add(s) and the LayerNorm of the NEXT half-layer that reads the sum, bit for bit the two calls.  On the register-stationary
 * route with 512 output columns the second product runs one row tile a workgroup (igemm_ask_kernel) and normalises in its epilogue. */
int lele_hip_fused_ffn_quantized_ln(LeleCtx* ctx, const LeleTensor* input, const LeleTensor* w1_int8, const LeleTensor* w1_scale,
                                    const LeleTensor* w1_zero, const LeleTensor* b1, const LeleTensor* w2_int8, const LeleTensor* w2_scale,
                                    const LeleTensor* w2_zero, const LeleTensor* b2, int apply_relu2, const LeleTensor* res1,
                                    const LeleTensor* res2, const LeleTensor* ln_scale, const LeleTensor* ln_bias, float epsilon, LeleBuf* out,
                                    LeleBuf* ln_out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && input && ln_scale && ln_bias && out && ln_out, "fused_ffn_quantized_ln: NULL argument");
    LELE_REQUIRE(out != ln_out, "fused_ffn_quantized_ln: the sum and its normalised form need two buffers");
    LnReq ln{ln_scale, ln_bias, epsilon, ln_out};
    int64_t sh[LELE_MAX_RANK];
    int32_t r = 0;
    LELE_TRY(ffn_impl(ctx, input, w1_int8, w1_scale, w1_zero, b1, w2_int8, w2_scale, w2_zero, b2, apply_relu2, res1, res2, out, sh, &r, &ln));
    if (!ln.done) {
        LeleTensor y{out->data, sh, r, LELE_F32, LELE_MEM_DEVICE};
        int64_t sh2[LELE_MAX_RANK];
        int32_t r2 = 0;
        LELE_TRY(lele_hip_layer_norm(ctx, &y, ln_scale, ln_bias, -1, epsilon, ln_out, sh2, &r2));
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(sh, sh + r));
}

/* ---- per-stage stopwatch of fused_quantized_linear (bench.py's roofline block for the model path) ------------------- */
int lele_hip_quant_set_profiling(LeleCtx* ctx, int on) {
    LELE_REQUIRE(ctx, "quant_set_profiling: ctx is NULL");
    ctx->qprof.on = on != 0;
    ctx->qprof.used = 0;
    return 0;
}
int lele_hip_quant_profile_read(LeleCtx* ctx, float* range_ms, float* quantise_ms, float* gemm_ms, int64_t* calls) {
    LELE_REQUIRE(ctx && range_ms && quantise_ms && gemm_ms && calls, "quant_profile_read: NULL argument");
    LELE_REQUIRE(!ctx->capturing, "quant_profile_read: not allowed while a graph is being captured");
    LELE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double t[3] = {0, 0, 0};
    auto& p = ctx->qprof;
    const size_t n = p.used / 4;
    for (size_t i = 0; i < n; ++i)
        for (int st = 0; st < 3; ++st) {
            float ms = 0.0f;
            LELE_HIP_CHECK(hipEventElapsedTime(&ms, p.ev[4 * i + st], p.ev[4 * i + st + 1]));
            t[st] += ms;
        }
    *calls = (int64_t)n;
    *range_ms = n ? (float)(t[0] / n) : 0.0f;
    *quantise_ms = n ? (float)(t[1] / n) : 0.0f;
    *gemm_ms = n ? (float)(t[2] / n) : 0.0f;
    p.used = 0;
    return 0;
}

/* ---- prepared weights: prepare_weights (quantization.rs:221), mat_mul_integer_prepared (699), fused_dq_gemm_prepared_x86
 *      (454), mat_mul_integer_u8_weights (173).  The handle owns the device copy of the RAW u8 matrix carried as f32 (the
 *      encoding every other entry point takes) and marks it immutable, so the packed forms are built once and cached. ---- */
struct LelePrepared {
    LeleCtx* ctx;
    std::vector<float> host;  // u8 values as f32 [k, n]: the LELE_MEM_WEIGHT identity of this matrix
    int64_t k, n;
    int64_t shape[2];
};
int lele_hip_prepare_weights(LeleCtx* ctx, const uint8_t* b_u8, int64_t k, int64_t n, LelePrepared** out) {
    LELE_REQUIRE(ctx && b_u8 && out, "prepare_weights: NULL argument");
    LELE_REQUIRE(k > 0 && n > 0, "prepare_weights: empty matrix (k=%lld, n=%lld)", (long long)k, (long long)n);
    LelePrepared* p = new LelePrepared();
    p->ctx = ctx;
    p->k = k;
    p->n = n;
    p->shape[0] = k;
    p->shape[1] = n;
    p->host.resize((size_t)k * n);
    for (int64_t i = 0; i < k * n; ++i) p->host[i] = (float)b_u8[i];
    *out = p;
    return 0;
}
int lele_hip_prepared_destroy(LelePrepared* p) {
    if (!p) return 0;
    // the cached device copies stay with the ctx (freed at ctx_destroy); drop the keys so a recycled host address cannot alias
    LeleCtx* c = p->ctx;
    (void)hipStreamSynchronize(c->stream);
    for (int tag : {0, 201, 202, 203, 204}) {
        auto it = c->weights.find(std::make_tuple((const void*)p->host.data(), p->host.size() * 4, tag));
        if (it != c->weights.end()) {
            (void)hipFree(it->second);
            ++c->generation;
            c->weights.erase(it);
        }
    }
    delete p;
    return 0;
}
static LeleTensor prepared_tensor(const LelePrepared* p) {
    return LeleTensor{p->host.data(), p->shape, 2, LELE_F32, LELE_MEM_WEIGHT};
}
int lele_hip_mat_mul_integer_prepared(LeleCtx* ctx, const LeleTensor* a, const LelePrepared* pw, int has_a_zero_point,
                                      float a_zero_point, int has_b_zero_point, int32_t b_zero_point, const LeleTensor* scale,
                                      const LeleTensor* bias, int apply_relu, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && pw && out, "mat_mul_integer_prepared: NULL argument");
    const LeleTensor b = prepared_tensor(pw);
    const float za = has_a_zero_point ? a_zero_point : 0.0f, zb = has_b_zero_point ? (float)(b_zero_point & 0xff) : 0.0f;
    const int64_t one = 1;
    const LeleTensor tza{&za, &one, 1, LELE_F32, LELE_MEM_HOST}, tzb{&zb, &one, 1, LELE_F32, LELE_MEM_HOST};
    return lele_hip_mat_mul_integer_with_scale_bias(ctx, a, &b, &tza, &tzb, scale, bias, apply_relu, out, out_shape, out_rank);
}
int lele_hip_fused_dq_gemm_prepared(LeleCtx* ctx, const LeleTensor* input, const LelePrepared* pw, int has_b_zero_point,
                                    int32_t b_zero_point, const LeleTensor* weight_scale, const LeleTensor* bias, int apply_relu,
                                    LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && input && pw && weight_scale && out, "fused_dq_gemm_prepared: NULL argument");
    const LeleTensor b = prepared_tensor(pw);
    const float zb = has_b_zero_point ? (float)(b_zero_point & 0xff) : 0.0f;
    const int64_t one = 1;
    const LeleTensor tzb{&zb, &one, 1, LELE_F32, LELE_MEM_HOST};
    return lele_hip_fused_quantized_linear(ctx, input, &b, weight_scale, &tzb, bias, apply_relu, out, out_shape, out_rank);
}

int lele_hip_dynamic_quantize_linear(LeleCtx* ctx, const LeleTensor* x, LeleBuf* out_y, LeleBuf* out_scale,
                                     LeleBuf* out_zp, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out_y && out_scale && out_zp, "dynamic_quantize_linear: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t len = numel(x);
    std::vector<int64_t> shp(x->shape, x->shape + x->rank);
    LELE_TRY(out_y->reserve((size_t)len * 4));
    LELE_TRY(out_scale->reserve(4));
    LELE_TRY(out_zp->reserve(4));
    if (len == 0) {  // avx/quantization.rs:845-851: scale 1.0, zero point 0.0
        const float one = 1.0f, zero = 0.0f;
        LELE_HIP_CHECK(hipMemcpyAsync(out_scale->data, &one, 4, hipMemcpyHostToDevice, ctx->stream));
        LELE_HIP_CHECK(hipMemcpyAsync(out_zp->data, &zero, 4, hipMemcpyHostToDevice, ctx->stream));
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return set_shape_v(out_shape, out_rank, shp);
    }
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    void* prm = nullptr;
    LELE_TRY(ctx->arena_alloc(sizeof(QParams), &prm));
    LELE_TRY(launch_range(ctx, (const float*)dx, 1, len, (QParams*)prm, (float*)out_scale->data, (float*)out_zp->data));
    const int blocks = (int)std::min<int64_t>((len + 255) / 256, 2048);
    hipLaunchKernelGGL(dq_apply_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const float*)dx, len,
                       (const QParams*)prm, (float*)out_y->data);
    LELE_HIP_CHECK(hipGetLastError());
    return set_shape_v(out_shape, out_rank, shp);
}

int lele_hip_mat_mul_integer_with_scale_bias(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b,
                                             const LeleTensor* a_zero_point, const LeleTensor* b_zero_point,
                                             const LeleTensor* scale, const LeleTensor* bias, int apply_relu,
                                             LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && b && out, "mat_mul_integer: NULL argument");
    LELE_REQUIRE(a->rank >= 2 && b->rank >= 2, "mat_mul_integer: rank >= 2 required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t m = a->shape[a->rank - 2], k = a->shape[a->rank - 1];
    const int64_t kb = b->shape[b->rank - 2], n = b->shape[b->rank - 1];
    LELE_REQUIRE(k == kb, "mat_mul_integer: K mismatch (%lld vs %lld)", (long long)k, (long long)kb);
    int64_t batch_a = 1, batch_b = 1;
    for (int i = 0; i + 2 < a->rank; ++i) batch_a *= a->shape[i];
    for (int i = 0; i + 2 < b->rank; ++i) batch_b *= b->shape[i];
    LELE_REQUIRE(batch_a == batch_b || batch_a == 1 || batch_b == 1, "mat_mul_integer: batch %lld vs %lld",
                 (long long)batch_a, (long long)batch_b);
    const int64_t fb = std::max(batch_a, batch_b);
    std::vector<int64_t> shp;
    if (batch_a >= batch_b)
        shp.assign(a->shape, a->shape + a->rank - 2);
    else
        shp.assign(b->shape, b->shape + b->rank - 2);
    shp.push_back(m);
    shp.push_back(n);
    LELE_TRY(out->reserve((size_t)fb * m * n * 4));
    if (fb * m * n == 0) return set_shape_v(out_shape, out_rank, shp);
    auto first_as_int = [&](const LeleTensor* t, int* v) -> int {  // `.data.first().map(|&v| v as i32).unwrap_or(0)`
        *v = 0;
        if (!t || numel(t) == 0) return 0;
        float f = 0.0f;
        if (t->mem == LELE_MEM_DEVICE) {
            LELE_HIP_CHECK(hipMemcpyAsync(&f, t->data, 4, hipMemcpyDeviceToHost, ctx->stream));
            LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
            LELE_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        } else {
            f = *(const float*)t->data;
        }
        *v = (int)f;
        return 0;
    };
    int zp_a = 0, zp_b = 0;
    LELE_TRY(first_as_int(a_zero_point, &zp_a));
    LELE_TRY(first_as_int(b_zero_point, &zp_b));
    const int64_t slen = scale ? numel(scale) : 0, blen = bias ? numel(bias) : 0;
    LELE_REQUIRE(!scale || slen == 1 || slen >= n, "mat_mul_integer: scale has %lld entries for N=%lld",
                 (long long)slen, (long long)n);
    LELE_REQUIRE(!bias || blen >= n, "mat_mul_integer: bias has %lld entries for N=%lld", (long long)blen, (long long)n);
    LELE_TRY(ctx->arena_reset());
    const void *da = nullptr, *db = nullptr, *ds = nullptr, *dbi = nullptr;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    if (scale) LELE_TRY(ctx->dev_ptr(scale, &ds));
    if (bias) LELE_TRY(ctx->dev_ptr(bias, &dbi));
    const int kp = (int)((k + 15) & ~int64_t(15));
    PackedW pw;
    LELE_TRY(get_packed_weights(ctx, b, (const float*)db, batch_b, (int)k, (int)n, kp, &pw,
                                b->mem == LELE_MEM_WEIGHT));
    const int64_t rows_a = batch_a * m;
    void *aq = nullptr, *rs = nullptr;
    LELE_TRY(ctx->arena_alloc((size_t)rows_a * kp, &aq));
    LELE_TRY(ctx->arena_alloc((size_t)rows_a * 4, &rs));
    hipLaunchKernelGGL(qrows_kernel<1>, dim3((unsigned)((rows_a + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)da,
                       rows_a, (int)k, kp, (int)m, (QParams*)nullptr, (int8_t*)aq, (int*)rs, (const float*)nullptr, 0, (unsigned*)nullptr, (int*)nullptr);
    // one launch per batch slice unless everything is un-batched on the B side (then all rows share the weights)
    const int64_t launches = (batch_b == 1 && batch_a >= 1) ? 1 : fb;
    for (int64_t bi = 0; bi < launches; ++bi) {
        const int64_t rows = launches == 1 ? rows_a : m;
        const int8_t* aptr = (const int8_t*)aq + (launches == 1 || batch_a == 1 ? 0 : bi * m * kp);
        const int* rsp = (const int*)rs + (launches == 1 || batch_a == 1 ? 0 : bi * m);
        IgemmEpi epi{(float*)out->data + (launches == 1 ? 0 : bi * m * n), rows, n, (int)m, (int)k, rsp,
                     pw.col_sums + (batch_b == 1 ? 0 : bi * n), nullptr, zp_a, zp_b, (const float*)ds, (int)slen,
                     (const float*)dbi, apply_relu};
        LELE_TRY(launch_igemm(ctx, aptr, pw.wt + (batch_b == 1 ? 0 : bi * n * kp), rows, (int)n, kp, 0, (int)m, epi));
    }
    return set_shape_v(out_shape, out_rank, shp);
}

}  // extern "C"
