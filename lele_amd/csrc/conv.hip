// conv.hip -- Conv2d / Conv1d / ConvTranspose on gfx950.
//
//   lele_hip_conv2d (act = none | relu | silu)  <- /root/reference/src/kernels/conv2d.rs:107-888 (conv2d, conv2d_fused,
//                                                   conv2d_silu -> conv2d_activation), im2col 892-1046, depthwise 3131-3384
//   lele_hip_conv1d (relu flag)                 <- /root/reference/src/kernels/conv1d.rs:837-1464 (conv1d, conv1d_fused)
//   lele_hip_conv_transpose                     <- /root/reference/src/kernels/conv2d.rs:2952-3128
//
// lele lowers convolution to im2col + faer GEMM (or direct AVX2 loops) and then runs a per-channel bias/activation
// pass.  Here the convolution is ONE implicit GEMM on the f32 MFMA core (gemm_core.h): M = OC/g, N = OH*OW,
// K = IC/g*kh*kw, batch = images * groups; the B operand is gathered straight from the NCHW input (no materialised
// im2col buffer) and bias + activation are applied in the epilogue while the accumulators are still in registers.
// Depthwise (IC/g == OC/g == 1) has no GEMM shape at all and runs as a direct, HBM-bound stencil kernel.
// Semantics are ONNX's (the reference's own in-test oracle ref_conv2d); the x86 path's documented defects
// (depthwise bias/SiLU dropped, right-edge over-read) are NOT reproduced -- see DESIGN.md.
// Activation matches the x86 epilogue: polynomial SiLU for the first plane&~7 positions of each (n, oc) plane,
// libm form for the tail (avx/math.rs:344-365).
#include "common.h"
#include "gemm_core.h"
#include "simd_math.h"

#include <math.h>
#include <string.h>
#include <stdlib.h>

using namespace lele;

namespace {

// SiLU of a convolution's epilogue.  Default: x * r with r = 1 / (1 + exp2(-x * log2 e)) from the transcendental unit (v_exp_f32,
// v_rcp_f32) and ONE Newton step on the reciprocal (two FMAs: v_rcp_f32 alone leaves a result that is 8e-9 low on average -- nothing
// for one layer, but a bias is what a deep network adds up coherently: profiles/r05_graph_error_growth.json): 7 instructions; within
// 1e-5 relative + 1e-7 of the reference's form over the whole f32 range (tests/test_conv_rnn.py), mean error 2e-9 -- the sum it is
// applied to is itself pinned to 1e-4 only (the summation order of faer is not), and the replica of the reference's epilogue
// (avx/math.rs: 24 instructions a value: a degree-7 polynomial exp, a Newton-refined reciprocal; libm exp and a division on the last
// 0-7 positions of a plane) was 10 % of a Yolo-shaped forward at batch 64.  LELE_HIP_CONV_SILU_EXACT=1 selects the replica
// (kActSiluExact; conv2d_entry).  The stand-alone silu / sigmoid operators and the ConvInteger epilogues are replicas always.
constexpr int kActSiluExact = 3;
__device__ __forceinline__ float silu_fast(float v) {
    const float d = 1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896341f);
    float r = __builtin_amdgcn_rcpf(d);
    r = __builtin_fmaf(__builtin_fmaf(-d, r, 1.0f), r, r);  // d = inf (x < -88): r = 0, fma(-inf, 0, 1) = NaN -> guarded below
    return d < 3.0e38f ? v * r : 0.0f * v;                   // x * 0 keeps the sign of zero / NaN of the plain form
}
__device__ __forceinline__ float apply_act(float v, int act, bool body) {
    if (act == LELE_ACT_RELU) return v > 0.0f ? v : 0.0f;
    if (act == LELE_ACT_SILU) return silu_fast(v);
    if (act == kActSiluExact) return body ? silu_poly(v) : v / (1.0f + expf(-v));
    return v;
}

// Developer's knock-out builds of the window kernels (tools/conv_ko_build.sh: -DLELE_CONV_KO=<bits>, never the product): 2 = next to
// nothing is stored, 4 = one product instead of six, 8 = the window loads read past their resource (zeros, no memory access).
#ifndef LELE_CONV_KO
#define LELE_CONV_KO 0
#endif
struct ConvGeom {
    int n, c, ih, iw, oc, kh, kw, group, icg, ocg, pt, pl, sh, sw, dh, dw, oh, ow, K, plane;
    // elements from one image to the next in x / out.  0 = dense (c*ih*iw / oc*plane, filled in by run_conv2d); anything else is a
    // CHANNEL VIEW of a wider NCHW tensor (lele_hip_conv2d_pitched: a Concat operand written in place, a Split result read in place)
    long long xbs = 0, obs = 0;
    // a residual [n, oc, oh, ow] (rbs elements from image to image) added AFTER the activation: out = act(conv + bias) + res, the
    // bits of the convolution followed by a separate Add (lele_hip_conv2d_res)
    const float* res = nullptr;
    long long rbs = 0;
};

// exact n / d for the small non-negative values of this file via one mulhi (m = floor(2^32/d) + 1 is exact while
// n * d < 2^32, which the host checks; otherwise use_magic is 0 and a real division is used)
struct FastDiv {
    unsigned m, d;
    int use_magic;
    __device__ __forceinline__ int div(int n) const { return use_magic ? (int)__umulhi((unsigned)n, m) : n / (int)d; }
};
inline FastDiv make_fastdiv(int d, int64_t max_n) {
    FastDiv f;
    f.d = (unsigned)d;
    f.m = (unsigned)((uint64_t(1) << 32) / (unsigned)d + 1);
    f.use_magic = d > 1 && max_n * d < (int64_t(1) << 32);
    if (d == 1) {  // n / 1: the magic would overflow
        f.use_magic = 0;
    }
    return f;
}

// A operand: weights [OC][ICg*kh*kw] row-major; GEMM batch b = img*G + g selects the group's rows
// (loader protocol: gemm_core.h -- row() is the k-invariant part, hoisted out of the K loop by the kernel)
struct ConvWLoad {
    const float* w;
    ConvGeom g;
    int vec;
    static constexpr bool kRowFast = false;
    struct Row {
        const float* q;
        bool rin;
    };
    __device__ __forceinline__ Row row(int b, int r) const {
        const bool rin = r < g.ocg;
        return Row{w + ((int64_t)((b % g.group) * g.ocg + (rin ? r : g.ocg - 1))) * g.K, rin};
    }
    __device__ __forceinline__ float4 get4(const Row& r, int k) const {
        float4 v;
        if (vec && k + 3 < g.K) {
            v = *reinterpret_cast<const float4*>(r.q + k);
        } else {
            const int last = g.K - 1;
            const float e0 = r.q[k + 0 < g.K ? k + 0 : last], e1 = r.q[k + 1 < g.K ? k + 1 : last];
            const float e2 = r.q[k + 2 < g.K ? k + 2 : last], e3 = r.q[k + 3 < g.K ? k + 3 : last];
            v = make_float4(k + 0 < g.K ? e0 : 0.f, k + 1 < g.K ? e1 : 0.f, k + 2 < g.K ? e2 : 0.f, k + 3 < g.K ? e3 : 0.f);
        }
        return r.rin ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ float4 get4(int b, int r, int k) const { return get4(row(b, r), k); }
};
// B operand: element(position p, k) = x[img][g*ICg + ic][oh*sh - pt + a*dh][ow*sw - pl + bb*dw] (0 outside).
// Index arithmetic only (mulhi divisions); the four loads are unconditional on clamped addresses.  The output position
// -> window origin arithmetic is per row (hoisted); the k -> (ic, a, bb) split is uniform over a wave in the tiled kernel.
struct ConvXRow {
    const float* base;  // x + (img, group) offset: uniform over the workgroup
    int iy0, ix0;
    bool rin;
};
struct ConvXLoad {
    const float* x;
    ConvGeom g;
    FastDiv d_ow, d_khw, d_kw;
    static constexpr bool kRowFast = true;  // consecutive threads -> consecutive output positions (coalesced along W)
    typedef ConvXRow Row;
    __device__ __forceinline__ Row row(int b, int r) const {
        const bool rin = r < g.plane;
        const int rowc = rin ? r : g.plane - 1;
        const int img = b / g.group, grp = b - img * g.group;
        const int oy = d_ow.div(rowc), ox = rowc - oy * g.ow;
        return Row{x + (int64_t)img * g.xbs + (int64_t)grp * g.icg * g.ih * g.iw, oy * g.sh - g.pt, ox * g.sw - g.pl, rin};
    }
    __device__ __forceinline__ float4 get4(const Row& r, int k) const {
        const int kc = k < g.K ? k : g.K - 1;
        int ic = d_khw.div(kc), rem = kc - ic * (g.kh * g.kw);
        int a = d_kw.div(rem), bb = rem - a * g.kw;
        int idx[4];
        bool ok[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int iy = r.iy0 + a * g.dh, ix = r.ix0 + bb * g.dw;
            ok[e] = r.rin && k + e < g.K && iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw;
            idx[e] = ok[e] ? (ic * g.ih + iy) * g.iw + ix : 0;
            if (++bb == g.kw) {
                bb = 0;
                if (++a == g.kh) {
                    a = 0;
                    ++ic;
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) LELE_DEV_ASSERT(idx[e] >= 0 && idx[e] < g.icg * g.ih * g.iw);  // inside this image group's planes
        const float e0 = r.base[idx[0]], e1 = r.base[idx[1]], e2 = r.base[idx[2]], e3 = r.base[idx[3]];
        return make_float4(ok[0] ? e0 : 0.f, ok[1] ? e1 : 0.f, ok[2] ? e2 : 0.f, ok[3] ? e3 : 0.f);
    }
    __device__ __forceinline__ float4 get4(int b, int r, int k) const { return get4(row(b, r), k); }
};
// Tap-major variant (IC/g % 4 == 0): the GEMM K index is (tap, ic) instead of lele's (ic, tap) -- a permutation of the
// exact f32 sum -- so the 4 consecutive k of a chunk are 4 input channels at ONE tap: one bounds test and one base
// index per chunk, four loads at the constant stride ih*iw.  The weights are re-laid out once to [OC][tap][ic]
// (cached for LELE_MEM_WEIGHT tensors).
struct ConvXLoadTap {
    const float* x;
    ConvGeom g;
    FastDiv d_ow, d_icg, d_kw;
    static constexpr bool kRowFast = true;
    typedef ConvXRow Row;
    __device__ __forceinline__ Row row(int b, int r) const {
        const bool rin = r < g.plane;
        const int rowc = rin ? r : g.plane - 1;
        const int img = b / g.group, grp = b - img * g.group;
        const int oy = d_ow.div(rowc), ox = rowc - oy * g.ow;
        return Row{x + (int64_t)img * g.xbs + (int64_t)grp * g.icg * g.ih * g.iw, oy * g.sh - g.pt, ox * g.sw - g.pl, rin};
    }
    __device__ __forceinline__ float4 get4(const Row& r, int k) const {
        const int kc = k < g.K ? k : g.K - 4;  // K % 4 == 0 here, so a chunk is entirely in or out of range
        const int tap = d_icg.div(kc), ic = kc - tap * g.icg;
        const int a = d_kw.div(tap), bb = tap - a * g.kw;
        const int iy = r.iy0 + a * g.dh, ix = r.ix0 + bb * g.dw;
        const bool ok = r.rin && k < g.K && iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw;
        const int hw = g.ih * g.iw;
        // (base + ic*hw) is uniform when k is; the lane adds its 32-bit position inside the plane
        const float* q = r.base + (int64_t)ic * hw;
        const unsigned off = ok ? (unsigned)(iy * g.iw + ix) : 0u;
        LELE_DEV_ASSERT(ic >= 0 && ic + 3 < g.icg && off < (unsigned)hw);  // four channel planes of this image group
        const float e0 = q[off], e1 = (q + hw)[off], e2 = (q + 2 * hw)[off], e3 = (q + 3 * hw)[off];
        return ok ? make_float4(e0, e1, e2, e3) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __device__ __forceinline__ float4 get4(int b, int r, int k) const { return get4(row(b, r), k); }
};
__global__ void conv_wperm_kernel(const float* __restrict__ w, float* __restrict__ wt, int oc, int icg, int khw) {
    const int64_t total = (int64_t)oc * icg * khw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ic = (int)(i % icg), tap = (int)((i / icg) % khw);
        const int64_t o = i / ((int64_t)icg * khw);
        wt[i] = w[(o * icg + ic) * khw + tap];
    }
}

struct ConvEpi {
    float* out;
    const float* bias;
    ConvGeom g;
    int act;
#ifdef LELE_HIP_LAB
    long long* dbg = nullptr;  // lab switch LELE_HIP_CONV_STAMPS (tools/conv_stamps.py): [workgroup][8 waves][64] shader-clock stamps of the window kernel
#endif
    __device__ __forceinline__ float load(int b, int row, int col) const {  // clamped coordinates, unconditional
        return bias ? bias[(b % g.group) * g.ocg + row] : 0.0f;
    }
    // the activation of one value: the SiLU's scalar-tail form (libm exp, a division) only when some active lane sits in the last
    // 0-7 positions of its plane -- evaluated per lane behind a select, both forms ran for every element of every tile
    __device__ __forceinline__ float activate(float v, int col) const {
        if (act == LELE_ACT_NONE) return v;
        if (act == LELE_ACT_RELU) return v > 0.0f ? v : 0.0f;
        if (act == LELE_ACT_SILU) return silu_fast(v);
        const bool body = col < (g.plane & ~7);
        if (__builtin_amdgcn_ballot_w64(!body) == 0) return apply_act(v, kActSiluExact, true);
        return apply_act(v, kActSiluExact, body);
    }
    __device__ __forceinline__ void store(int b, int row, int col, float acc, float pre) const {
        if (row >= g.ocg || col >= g.plane) return;
        const int img = b / g.group, o = (b % g.group) * g.ocg + row;
        float v = acc;
        if (bias) v = v + pre;
        v = activate(v, col);
        LELE_DEV_ASSERT(img >= 0 && img < g.n && o >= 0 && o < g.oc && col >= 0 && col < g.plane);
        if (g.res) v = v + g.res[(int64_t)img * g.rbs + (int64_t)o * g.plane + col];
        out[(int64_t)img * g.obs + (int64_t)o * g.plane + col] = v;
    }
    // the 16-byte store protocol of gemm_core.h: finished values, a row's address, and when rows may be written four columns at a time
    __device__ __forceinline__ bool vec_ok() const {
        return (g.plane & 3) == 0 && (g.obs & 3) == 0 && (((uintptr_t)out) & 15) == 0 && (!g.res || ((g.rbs & 3) == 0 && (((uintptr_t)g.res) & 15) == 0));
    }
    __device__ __forceinline__ float finish(int b, int row, int col, float acc, float pre) const {  // clamped coordinates
        float v = acc;
        if (bias) v = v + pre;
        return activate(v, col);
    }
    __device__ __forceinline__ float* row_ptr(int b, int row) const {
        const int img = b / g.group, o = (b % g.group) * g.ocg + row;
        return out + (int64_t)img * g.obs + (int64_t)o * g.plane;
    }
    // the residual's row beside a row of finished values (gemm_core.h adds it to the 16-byte pieces it stores), or NULL
    __device__ __forceinline__ const float* res_row_ptr(int b, int row) const {
        const int img = b / g.group, o = (b % g.group) * g.ocg + row;
        return g.res ? g.res + (int64_t)img * g.rbs + (int64_t)o * g.plane : nullptr;
    }
};

// depthwise: one thread per output element, taps in (kh, kw) order, FMA chain in f32.  32-bit index arithmetic (the
// host checks the element count), grid-stride over the flattened (plane, position) space so that small planes still
// give full workgroups; every tap is an unconditional load from a clamped address followed by a select (a
// bounds-checked load would serialise one memory round trip per tap).
__global__ __launch_bounds__(256) void depthwise_conv2d_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ out,
                                                               ConvGeom g, int act, unsigned total) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned pl = i / (unsigned)g.plane;
        const int p = (int)(i - pl * (unsigned)g.plane);
        const int oy = p / g.ow, ox = p - oy * g.ow;
        const int iy0 = oy * g.sh - g.pt, ix0 = ox * g.sw - g.pl;
        const int ch = (int)(pl % (unsigned)g.oc);
        const float* xp = x + (int64_t)pl * g.ih * g.iw;  // depthwise: input plane index == output plane index
        const float* wp = w + (int64_t)ch * g.kh * g.kw;
        float acc = 0.0f;
        for (int a = 0; a < g.kh; ++a) {
            const int iy = iy0 + a * g.dh;
            const bool yin = iy >= 0 && iy < g.ih;
            const int rowoff = (yin ? iy : 0) * g.iw;
#pragma unroll 4
            for (int b = 0; b < g.kw; ++b) {
                const int ix = ix0 + b * g.dw;
                const bool in = yin && ix >= 0 && ix < g.iw;
                const float xv = xp[rowoff + (in ? ix : 0)];
                const float wv = wp[a * g.kw + b];
                acc = in ? fmaf_(xv, wv, acc) : acc;  // skipped taps leave the chain untouched, as the reference's loop does
            }
        }
        if (bias) acc = acc + bias[ch];
        out[i] = apply_act(acc, act, p < (g.plane & ~7));
    }
}

// depthwise with unit stride and dilation along the width: one thread produces FOUR consecutive outputs of a row from
// a sliding window of kw + 3 inputs held in registers, and every weight is fetched once per thread -- (kw + 3) + kw
// loads per tap row instead of 8 kw.  (The one-output-per-thread kernel above is bound by the rate at which the texture
// path accepts wave-wide 4-byte loads: 25 us for the [32, 512, 171] k = 11 FSMN convolution.)  Same tap order, same
// skip-when-out-of-range select, same FMA chain per output: bit-identical results.
struct DwGeom {  // the scalars the row kernel needs (the full ConvGeom spills SGPRs)
    int ih, iw, oh, ow, oc, kh, sh, dh, pt, pl, body;
};
template <int KW>
__global__ __launch_bounds__(256) void depthwise_row4_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, float* __restrict__ out,
                                                             DwGeom g, int act, unsigned total, unsigned ngroups) {
    const unsigned i = blockIdx.x * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned t = i / ngroups, gq = i - t * ngroups;
    const unsigned pl = t / (unsigned)g.oh;
    const int oy = (int)(t - pl * (unsigned)g.oh);
    const int ox0 = (int)gq * 4, ix0 = ox0 - g.pl, iy0 = oy * g.sh - g.pt;
    const int ch = (int)(pl % (unsigned)g.oc);
    const float* xp = x + (int64_t)pl * g.ih * g.iw;
    const float* wp = w + ch * g.kh * KW;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int a = 0; a < g.kh; ++a) {
        const int iy = iy0 + a * g.dh;
        const bool yin = iy >= 0 && iy < g.ih;
        const float* row = xp + (yin ? iy : 0) * g.iw;
        float xs[KW + 3], wv[KW];
#pragma unroll
        for (int j = 0; j < KW + 3; ++j) {
            const int ix = ix0 + j;
            xs[j] = row[min(max(ix, 0), g.iw - 1)];
        }
#pragma unroll
        for (int b = 0; b < KW; ++b) wv[b] = wp[a * KW + b];
#pragma unroll
        for (int j = 0; j < KW + 3; ++j) {  // an out-of-range input position is skipped by every output that taps it
            const int ix = ix0 + j;
            const bool in = yin && ix >= 0 && ix < g.iw;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = j - q;
                if (b >= 0 && b < KW) {
                    // accumulate in tap order per output: output q sees taps b = 0..KW-1 as j = q..q+KW-1 ascends
                    const float f = fmaf_(xs[j], wv[b], acc[q]);
                    acc[q] = in ? f : acc[q];
                }
            }
        }
    }
    const float bv = bias ? bias[ch] : 0.0f;
    float* orow = out + ((int64_t)pl * g.oh + oy) * g.ow;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ox = ox0 + q;
        if (ox < g.ow) {
            float v = acc[q];
            if (bias) v = v + bv;
            orow[ox] = apply_act(v, act, oy * g.ow + ox < g.body);
        }
    }
}

// LDS-staged form of the same computation for planes that fit in LDS: a workgroup copies PB whole input planes
// (one contiguous run of global memory, 16-byte loads when aligned) into LDS, every thread then slides its window over
// LDS, results are collected in LDS and leave as one contiguous run.  The row kernel above reads global memory with a
// 16-byte lane stride, which the vector L1 handles as one access per lane (rocprofv3: 45 cache accesses per load
// instruction, 16 us for 22 MB); this form issues only fully coalesced global accesses.  Arithmetic is identical.
__device__ __forceinline__ void dw_copy_run(const float* __restrict__ src, float* __restrict__ dst, unsigned count, bool vec) {
    if (vec) {  // src and dst 16-byte aligned
        const unsigned nv = count >> 2;
        for (unsigned e = threadIdx.x; e < nv; e += 256u)
            reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(src)[e];
        for (unsigned e = 4 * nv + threadIdx.x; e < count; e += 256u) dst[e] = src[e];
    } else {
        for (unsigned e = threadIdx.x; e < count; e += 256u) dst[e] = src[e];
    }
}
template <int KW>
__global__ __launch_bounds__(256) void depthwise_lds_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            DwGeom g, int act, unsigned planes, unsigned pb, unsigned ngroups) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];
    const unsigned ihw = (unsigned)(g.ih * g.iw), ohw = (unsigned)(g.oh * g.ow);
    const unsigned pl0 = blockIdx.x * pb, np = planes - pl0 < pb ? planes - pl0 : pb;
    float* lin = dw_lds;
    float* lout = dw_lds + ((pb * ihw + 3u) & ~3u);
    const float* src = x + (size_t)pl0 * ihw;
    float* dst = out + (size_t)pl0 * ohw;
    dw_copy_run(src, lin, np * ihw, (((uintptr_t)src) & 15) == 0);
    __syncthreads();
    const bool direct = (g.ow & 3) == 0 && (((uintptr_t)dst) & 15) == 0 && (ohw & 3u) == 0;   // uniform
    const unsigned per_plane = (unsigned)g.oh * ngroups, items = np * per_plane;
    for (unsigned it = threadIdx.x; it < items; it += 256u) {
        const unsigned p = it / per_plane, r = it - p * per_plane;
        const int oy = (int)(r / ngroups);
        const int ox0 = (int)(r - (unsigned)oy * ngroups) * 4, ix0 = ox0 - g.pl, iy0 = oy * g.sh - g.pt;
        const int ch = (int)((pl0 + p) % (unsigned)g.oc);
        const float* xp = lin + p * ihw;
        const float* wp = w + ch * g.kh * KW;
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int a = 0; a < g.kh; ++a) {
            const int iy = iy0 + a * g.dh;
            const bool yin = iy >= 0 && iy < g.ih;
            const float* row = xp + (yin ? iy : 0) * g.iw;
            float xs[KW + 3], wv[KW];
#pragma unroll
            for (int j = 0; j < KW + 3; ++j) xs[j] = row[min(max(ix0 + j, 0), g.iw - 1)];
#pragma unroll
            for (int b = 0; b < KW; ++b) wv[b] = wp[a * KW + b];
#pragma unroll
            for (int j = 0; j < KW + 3; ++j) {
                const int ix = ix0 + j;
                const bool in = yin && ix >= 0 && ix < g.iw;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int b = j - q;
                    if (b >= 0 && b < KW) {
                        const float f = fmaf_(xs[j], wv[b], acc[q]);
                        acc[q] = in ? f : acc[q];
                    }
                }
            }
        }
        const float bv = bias ? bias[ch] : 0.0f;
        float r4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v = acc[q];
            if (bias) v = v + bv;
            r4[q] = apply_act(v, act, oy * g.ow + ox0 + q < g.body);
        }
        // a thread's four outputs are consecutive in memory and so are the threads of a row: rows a multiple of four wide leave as
        // 16-byte stores straight from the registers (the second turn through LDS this kernel used to take cost a barrier and half
        // of its LDS); ragged rows go through LDS as before
        if (direct) {
            *reinterpret_cast<float4*>(dst + p * ohw + (unsigned)oy * (unsigned)g.ow + (unsigned)ox0) = make_float4(r4[0], r4[1], r4[2], r4[3]);
        } else {
            float* orow = lout + p * ohw + (unsigned)oy * (unsigned)g.ow;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (ox0 + q < g.ow) orow[ox0 + q] = r4[q];
        }
    }
    if (direct) return;
    __syncthreads();
    dw_copy_run(lout, dst, np * ohw, (((uintptr_t)dst) & 15) == 0);
}

// Depthwise 1-D convolution on a TIME-MAJOR tensor x[B][T][C] (channels innermost): what the three-node sequence
// Transpose(0,2,1) -> Conv(group = C, kernel k) -> Transpose(0,2,1) computes (an FSMN memory block exports that way),
// without the two transposes.  A lane owns one channel (coalesced 256-byte rows across the wave) and produces TT
// consecutive time steps from a sliding register window.  Per output: taps in ascending order, out-of-range taps skipped,
// FMA chain, bias added afterwards -- the arithmetic of the depthwise kernels above, so the result is bit-identical to
// the unfused sequence.
template <int KW, int TT>
__global__ __launch_bounds__(256) void dwconv1d_tlc_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, int t_in,
                                                           int t_out, int c, int pitch /* elements per time step of x */, int pl,
                                                           int relu, int add_input, unsigned tiles_t, unsigned total) {
    // Workgroup b runs on XCD b % 8, each with its own L2: in launch order the time tiles next to each other -- whose windows overlap
    // by KW - 1 rows -- sit on different XCDs and every one of them fetches the shared rows again (counters: 25 MB read for the
    // 11 MB of v on a configs[3] shard).  Re-labelled so that every XCD walks one contiguous range of (utterance, time tile, channels).
    const unsigned G = gridDim.x, xcd = blockIdx.x & 7u, gbase = G >> 3, grem = G & 7u;
    const unsigned logical = xcd * gbase + (xcd < grem ? xcd : grem) + (blockIdx.x >> 3);
    const unsigned i = logical * 256u + threadIdx.x;
    if (i >= total) return;
    const unsigned ch = i % (unsigned)c, r = i / (unsigned)c;
    const unsigned tile = r % tiles_t, b = r / tiles_t;
    const int t0 = (int)tile * TT;
    const float* xp = x + ((size_t)b * t_in) * pitch + ch;
    float xs[KW + TT - 1], wv[KW];
#pragma unroll
    for (int j = 0; j < KW + TT - 1; ++j) {
        const int t = t0 - pl + j;
        xs[j] = xp[(size_t)min(max(t, 0), t_in - 1) * pitch];
    }
#pragma unroll
    for (int j = 0; j < KW; ++j) wv[j] = w[ch * KW + j];
    const float bv = bias ? bias[ch] : 0.0f;
    float* op = out + ((size_t)b * t_out) * c + ch;
#pragma unroll
    for (int q = 0; q < TT; ++q) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < KW; ++j) {
            const int t = t0 + q - pl + j;
            const float f = fmaf_(xs[q + j], wv[j], acc);
            acc = (t >= 0 && t < t_in) ? f : acc;
        }
        if (bias) acc = acc + bv;
        if (relu) acc = acc > 0.0f ? acc : 0.0f;
        // the FSMN residual (memory + input): x[t0 + q] sits at window index q + pl (the host checks pl <= KW - 1);
        // a separate unrolled select keeps the index a compile-time constant per (q, pl) pair
        if (add_input) {
            float xv = 0.0f;
#pragma unroll
            for (int j = 0; j < KW; ++j) xv = (j == pl) ? xs[q + j] : xv;
            acc = acc + xv;
        }
        if (t0 + q < t_out) op[(size_t)(t0 + q) * c] = acc;
    }
}

// conv_transpose (group 1): gather form of the reference's GEMM + col2im scatter (conv2d.rs:3060-3126)
struct CtGeom {
    int n, c, ih, iw, oc, kh, kw, pt, pl, sh, sw, dh, dw, oh, ow;
};
__global__ void conv_transpose_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                      const float* __restrict__ bias, float* __restrict__ out, CtGeom g) {
    const int64_t total = (int64_t)g.n * g.oc * g.oh * g.ow;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ox = (int)(i % g.ow), oy = (int)((i / g.ow) % g.oh);
        const int o = (int)((i / ((int64_t)g.ow * g.oh)) % g.oc);
        const int64_t img = i / ((int64_t)g.ow * g.oh * g.oc);
        float acc = 0.0f;
        for (int a = 0; a < g.kh; ++a) {
            const int ty = oy + g.pt - a * g.dh;
            if (ty < 0 || ty % g.sh) continue;
            const int iy = ty / g.sh;
            if (iy >= g.ih) continue;
            for (int b = 0; b < g.kw; ++b) {
                const int tx = ox + g.pl - b * g.dw;
                if (tx < 0 || tx % g.sw) continue;
                const int ix = tx / g.sw;
                if (ix >= g.iw) continue;
                const float* xp = x + ((img * g.c) * g.ih + iy) * g.iw + ix;
                const float* wp = w + ((int64_t)o * g.kh + a) * g.kw + b;
                for (int ci = 0; ci < g.c; ++ci)
                    acc = fmaf_(xp[(int64_t)ci * g.ih * g.iw], wp[(int64_t)ci * g.oc * g.kh * g.kw], acc);
            }
        }
        if (bias) acc = acc + bias[o];
        out[i] = acc;
    }
}

// conv_transpose as one implicit GEMM per output phase (oy mod sh, ox mod sw): within a phase the contributing taps
// are a fixed arithmetic progression a = a0 + t*sa (b likewise) and the input row is iy = j + off_a0 - t*da, i.e. a
// stride-1 convolution with "dilation" -da over the phase's sub-grid.  The k2/s2 upsampling of the YOLO neck is four
// 1x1 convolutions.  Weights are gathered once per phase into [OC][K_phase] (tap-major when C % 4 == 0).
struct ConvTEpi {
    float* out;
    const float* bias;
    FastDiv d_ni;
    int oc, oh, ow, py, px, sh, sw, ni, plane;
    __device__ __forceinline__ float load(int b, int row, int col) const { return bias ? bias[row] : 0.0f; }
    __device__ __forceinline__ void store(int b, int row, int col, float acc, float pre) const {
        if (row >= oc || col >= plane) return;
        const int j = d_ni.div(col), i = col - j * ni;
        float v = acc;
        if (bias) v = v + pre;
        out[(((int64_t)b * oc + row) * oh + (py + sh * j)) * ow + (px + sw * i)] = v;
    }
};
// conv_transpose whose kernel equals its stride (no padding, no dilation): every output has exactly ONE tap, so the whole operator
// is one GEMM [OC*kh*kw, C] x [C, ih*iw] per image whose row (oc, a, b) and column (y, x) land at out[oc][sh*y + a][sw*x + b]
// (conv2d.rs:3060-3126 computes the same products and scatters them through col2im).  The k2 / s2 up-sampling of a segmentation
// head's prototype branch is this case; as four per-phase GEMMs its stores were 4-byte writes at an 8-byte stride.
struct ConvTKsEpi {
    float* out;
    const float* bias;
    FastDiv d_iw, d_taps, d_kw;
    int oc, oh, ow, kh, kw, iw, plane, taps;  // plane = ih * iw (GEMM columns), taps = kh * kw
    __device__ __forceinline__ float load(int b, int row, int col) const { return bias ? bias[d_taps.div(row)] : 0.0f; }
    __device__ __forceinline__ void store(int b, int row, int col, float acc, float pre) const {
        if (row >= oc * taps || col >= plane) return;
        const int o = d_taps.div(row), t = row - o * taps, a = d_kw.div(t), bb = t - a * kw;
        const int y = d_iw.div(col), x = col - y * iw;
        float v = acc;
        if (bias) v = v + pre;
        out[(((int64_t)b * oc + o) * oh + (kh * y + a)) * ow + (kw * x + bb)] = v;
    }
    // kernel 2 x 2: the four rows a lane holds for one column are the four outputs of one (oc, y, x): two 8-byte stores, adjacent
    // lanes adjacent in memory
    __device__ __forceinline__ bool quad_ok() const { return kh == 2 && kw == 2 && (((uintptr_t)out) & 7) == 0; }
    __device__ __forceinline__ void store_quad(int b, int row0, int col, const float* acc, const float* pre) const {
        if (row0 >= oc * 4 || col >= plane) return;
        const int o = row0 >> 2;
        const int y = d_iw.div(col), x = col - y * iw;
        float* p = out + (((int64_t)b * oc + o) * oh + 2 * y) * ow + 2 * x;
        float2 r0 = make_float2(acc[0], acc[1]), r1 = make_float2(acc[2], acc[3]);
        if (bias) {
            r0.x = r0.x + pre[0], r0.y = r0.y + pre[1];
            r1.x = r1.x + pre[2], r1.y = r1.y + pre[3];
        }
        *reinterpret_cast<float2*>(p) = r0;
        *reinterpret_cast<float2*>(p + ow) = r1;
    }
};
// w [C][OC][kh][kw] -> [OC*kh*kw][C]
__global__ void convt_wks_kernel(const float* __restrict__ w, float* __restrict__ wt, int c, int oc, int taps) {
    const int64_t total = (int64_t)c * oc * taps;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % c);
        const int64_t row = i / c;  // (o, tap)
        wt[i] = w[(int64_t)ci * oc * taps + row];
    }
}
__global__ void convt_wphase_kernel(const float* __restrict__ w, float* __restrict__ wt, int c, int oc, int kh, int kw,
                                    int a0, int sa, int na, int b0, int sb, int nb, int tap_major) {
    const int ntap = na * nb;
    const int64_t total = (int64_t)oc * c * ntap;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i % ((int64_t)c * ntap));
        const int o = (int)(i / ((int64_t)c * ntap));
        const int tap = tap_major ? kk / c : kk % ntap, ci = tap_major ? kk % c : kk / ntap;
        const int a = a0 + (tap / nb) * sa, b = b0 + (tap % nb) * sb;
        wt[i] = w[(((int64_t)ci * oc + o) * kh + a) * kw + b];
    }
}
__global__ void convt_fill_phase_kernel(float* __restrict__ out, const float* __restrict__ bias, int n, int oc, int oh,
                                        int ow, int py, int px, int sh, int sw, int nj, int ni) {
    const int64_t total = (int64_t)n * oc * nj * ni;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(t % ni), j = (int)((t / ni) % nj);
        const int64_t no = t / ((int64_t)ni * nj);
        out[(no * oh + (py + sh * j)) * ow + (px + sw * i)] = bias ? bias[no % oc] : 0.0f;
    }
}

// ---- 3 x 3 convolutions with FEW output channels (OC <= 16, IC <= 64: the stem and the narrow bottlenecks of a Yolo-shaped
// network) -- HBM-bound by their byte counts, and an order of magnitude off that bound as an implicit GEMM whose 32-row MFMA
// tile is mostly padding.  Direct form: a workgroup owns 8 x 32 output pixels of one image, stages their input window (all
// channels, halo included, zeros outside the image) in LDS once, and every thread accumulates ALL output channels of its pixel in
// registers: per (input channel, tap) one LDS read feeds OCB FMAs whose weights arrive as SCALAR operands (the address is
// wave-uniform: [ic][tap][OCB] re-laid once and cached like the tap-major weights), so the vector pipe does nothing but FMAs.
// Sum order: (ic, tap) ascending per output -- a permutation of the exact-product sum like every other kernel of this file.
template <int OCB, int S>
__global__ __launch_bounds__(256) void conv3x3_direct_kernel(const float* __restrict__ x, const float* __restrict__ wq /*[IC][9][OCB]*/,
                                                             const float* __restrict__ bias, float* __restrict__ out, ConvGeom g, int act,
                                                             int tiles_x, int icc /* channels per LDS pass */) {
    extern __shared__ __attribute__((aligned(16))) float c3_lds[];
    constexpr int TH = 8, TW = 32, PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3, PP = PW + 1;  // odd row pitch: no 2-way pattern on S = 2
    const int tid = threadIdx.x, ty = tid >> 5, tx = tid & 31;
    // (tile, image) from an XCD-contiguous relabelling of the launch order, as in win_items(): neighbouring tiles share an L2
    const unsigned G = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x, xcd = lin & 7u, gbase = G >> 3, grem = G & 7u;
    const unsigned logical = xcd * gbase + (xcd < grem ? xcd : grem) + (lin >> 3);
    const int tile = (int)(logical % gridDim.x), tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
    const int img = (int)(logical / gridDim.x);
    const int oy = tyi * TH + ty, ox = txi * TW + tx;
    const int iy0 = tyi * TH * S - g.pt, ix0 = txi * TW * S - g.pl;
    const float* xin = x + (int64_t)img * g.xbs;
    float acc[OCB];
#pragma unroll
    for (int o = 0; o < OCB; ++o) acc[o] = 0.0f;
    for (int c0 = 0; c0 < g.c; c0 += icc) {
        const int nc = g.c - c0 < icc ? g.c - c0 : icc;
        if (c0) __syncthreads();
        for (int i = tid; i < nc * PH * PW; i += 256) {
            const int ic = i / (PH * PW), r = i - ic * (PH * PW), py = r / PW, px = r - py * PW;
            const int iy = iy0 + py, ix = ix0 + px;
            const bool in = iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw;
            const float v = xin[((int64_t)(c0 + ic) * g.ih + (in ? iy : 0)) * g.iw + (in ? ix : 0)];
            c3_lds[(ic * PH + py) * PP + px] = in ? v : 0.0f;
        }
        __syncthreads();
        for (int ic = 0; ic < nc; ++ic) {
            const float* p = c3_lds + (ic * PH + ty * S) * PP + tx * S;
            float xv[9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) xv[3 * a + b] = p[a * PP + b];
            const float* wv = wq + (int64_t)(c0 + ic) * 9 * OCB;  // wave-uniform: scalar loads
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int o = 0; o < OCB; ++o) acc[o] = __builtin_fmaf(xv[t], wv[t * OCB + o], acc[o]);
        }
    }
    if (oy >= g.oh || ox >= g.ow) return;
    const int pos = oy * g.ow + ox;
    const bool body = pos < (g.plane & ~7);
    float* o0 = out + (int64_t)img * g.obs + pos;
#pragma unroll
    for (int o = 0; o < OCB; ++o)
        if (o < g.oc) {
            float v = acc[o];
            if (bias) v = v + bias[o];
            v = apply_act(v, act, body);
            if (g.res) v = v + g.res[(int64_t)img * g.rbs + (int64_t)o * g.plane + pos];  // lele_hip_conv2d_res
            o0[(int64_t)o * g.plane] = v;
        }
}
__global__ void conv3x3_wperm_kernel(const float* __restrict__ w, float* __restrict__ wq, int oc, int ic, int ocb) {
    const int total = ic * 9 * ocb;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int o = i % ocb, t = (i / ocb) % 9, c = i / (9 * ocb);
        wq[i] = o < oc ? w[((int64_t)o * ic + c) * 9 + t] : 0.0f;
    }
}

// ---- 3 x 3 (and large-plane 1 x 1), stride 1, IC % 16 == 0, over a batch: the input window is fetched ONCE -------------------
// As an implicit GEMM the nine taps of a 3 x 3 convolution read the same input window nine times through element-wise gathers,
// and the f32 MFMA shares the vector pipe with the gathers' address arithmetic: 0.45-0.5 of the f32 MFMA rate at 64-128
// channels.  Here a workgroup (4 consumer waves) owns 64 output channels x 256 positions of one image (WinTile) and walks the input channels
// in chunks of 16: the chunk's window (10 x 34 positions, zeros outside the image) is fetched once, every value cut into its
// three bf16 pieces on the way into LDS ([position][piece][16 channels]: a B fragment of tap (a, b) is three 16-byte reads at
// the shifted position), and all nine taps run from it -- 9 x 24 split-bf16 MFMAs per wave between two barriers, the next chunk's
// window in flight in registers meanwhile.  The weights are pre-split into MFMA fragment order once ([oc tile][chunk][tap][piece]
// [lane]: one coalesced 1 KiB load per piece, cached like the other re-laid weights).  Six-term products (gemm_core.h's note on
// split-bf16: exact-product sum to ~1e-7 per term, exact for integer-valued operands), f32 accumulation, ConvEpi's epilogue.
typedef __bf16 cbf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned cu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned cu32x2 __attribute__((ext_vector_type(2)));
typedef float cf32x16 __attribute__((ext_vector_type(16)));
constexpr int C3M_TH = 8, C3M_TW = 32, C3M_PITCH = 112;
template <int KS>
struct C3M {  // KS = 3 (3 x 3, padding in the geometry) or 1 (1 x 1: the "window" is the tile itself)
    static constexpr int PH = C3M_TH + KS - 1, PW = C3M_TW + KS - 1, POS = PH * PW, STAGE = POS * C3M_PITCH;
    static constexpr int TASKS = (POS * 4 + 255) / 256;  // (position, channel quad) staging tasks per thread and chunk
    static constexpr int TAPS = KS * KS;
};

// The tile of a workgroup of the window-once kernels: `th` rows x `tw` columns of output positions, th * tw <= the positions a workgroup
// multiplies (256 at stride 1, 128 at stride 2).  The MFMA's 32 columns are 32 CONSECUTIVE POSITIONS OF THE TILE IN ROW-MAJOR ORDER,
// not 32 columns of one row: position p sits at (p / tw, p % tw), so a 40-wide map is covered by 40 x 6 tiles (0.94 of the products
// land on outputs) and an 80-wide one by 16 x 16 tiles (1.0) where rows of 32 covered 0.62 and 0.83.  The host picks (tw, th) per
// layer (pick_win_tile); the window in LDS, the weight fragments and the products are the same for every shape.
struct WinTile {
    int tw, th, tiles_x;
    unsigned inv;  // ceil(2^20 / tw): p / tw == (p * inv) >> 20 for every p < 2^20 / tw
};
__device__ __forceinline__ int wt_row(const WinTile& t, int p) { return (int)(((unsigned)p * t.inv) >> 20); }


// Epilogue of the window-once kernels (c3m_epilogue_strips below): a consumer wave holds NJ accumulator tiles (32 output channels x
// NJ strips of 32 positions).  C layout: column = lane & 31 = the position inside the strip, rows (r & 3) + 8 (r >> 2) + 4 hv = output
// channels.  Stored as they sit, a lane would issue 16 NJ four-byte stores and the tile's tail is store-ISSUE bound (measured: 730 of
// 1370 us on 64 -> 64 channels at 160 x 160 x 64): the strips take a turn through LDS and leave as 16-byte pieces of 128-byte output
// rows.  Bias values are fetched together before anything else and the activation is chosen once per wave: written per element (a
// load behind `if (bias)` and a switch on the activation for each value) the compiler emitted 64 load -> wait -> polynomial chains
// one after the other, ~19 k cycles of a tile whose products take 28 k.
// ---- the stride-1 kernel: PERSISTENT workgroups of four loader waves and four consumer waves.
// Loaders ("producers", waves 4-7) and multipliers ("consumers", waves 0-3) are separate waves because a wave's memory counter is in
// order: a consumer with a window's loads in flight could not wait for its next weight fragment without waiting for the window too
// (938 us against 1529 for the tiled GEMM on 64 -> 64 channels at 160 x 160 x 64 when the kinds of wave were first separated).  The
// loaders fetch a chunk's window through buffer resources (a position outside the image is an offset past the resource: it reads as
// zero), keep two chunks in registers, and park one -- split into its bf16 pieces -- in the LDS stage the consumers have left.
// The workgroups are persistent because a workgroup that lives for one tile waits for its first window with nothing to overlap,
// multiplies for 0.3 us a chunk and stores its tile while its loaders have nothing left to fetch: measured with parts knocked out on
// 64 -> 64 channels 1 x 1 at 80 x 80 x 64 (69 us): no loads 55, no stores 35, neither 22, and 17.5 with no work at all -- loading and
// storing ADD UP, and launching 1600 short workgroups is a quarter of the time.  So at most two workgroups per CU are launched and
// each walks through its share of the items (image, tile, group of blocks of output channels), the blocks of an item in turn (the
// second block's window comes out of the same XCD's L2), as ONE stream of chunks: the loaders run two chunks ahead across item
// boundaries, so the next item's first windows are in flight while the consumers run the epilogue, and a wave's finished strip takes
// its turn through the stage the consumers have just left (32 x 32 values at a time: 20 KB for the four waves, inside one stage)
// while the other stage already holds the next chunk.  Barriers: B_q after chunk q is parked = before it is multiplied, plus E after
// an epilogue when a later chunk wants the stage the strips went through.  Whole Yolo-shaped forward at batch 64, same box,
// interleaved runs: 8.70-8.77 ms against 9.01-9.06 with one workgroup per tile.
template <int NJ, int OCT>
__device__ __forceinline__ void c3m_epilogue_strips(const cf32x16 (&acc)[NJ], const ConvEpi& epi, const ConvGeom& g, char* region, int wave, int lane,
                                                    int wm, int wn, int ocb, int img, int ty0, int tx0, const WinTile& wt
#ifdef LELE_HIP_LAB
                                                    , long long* dbg_w = nullptr, int* dbg_np = nullptr
#endif
) {
#ifdef LELE_HIP_LAB
#define E_STAMP()                                                                        \
    do {                                                                                 \
        if (dbg_np) {                                                                    \
            if (dbg_w && *dbg_np < 64 && (lane & 63) == 0) dbg_w[*dbg_np] = (long long)clock64(); \
            ++*dbg_np;                                                                   \
        }                                                                                \
    } while (0)
#else
#define E_STAMP()
#endif
    E_STAMP();  // e0
    // everything below that depends on the lane alone is cheap to compute and expensive to keep: left to itself the compiler hoists it
    // out of the persistent kernel's item loop, finds no registers for it beside the accumulators and the weight fragments, and reloads
    // it from scratch memory in every epilogue
    asm volatile("" : "+v"(lane));
    const int hv = lane >> 5, l31 = lane & 31;
    const int ocw = ocb * OCT + wm * 32;
    const int live = wt.tw * wt.th;
    if (g.ow % 4 == 0 && wt.tw % 4 == 0 && epi.vec_ok()) {
        // A strip (32 channels x 32 positions) goes through LDS AS IT IS and comes back as 16-byte pieces of output rows: a lane then
        // holds 4 consecutive positions of channel 8 it + (lane >> 3) -- bias, activation and residual are applied to those (the same
        // operations on the same values as before the turn), so the accumulators die as they are written and a lane needs 4 bias
        // values, not 16: the epilogue of a 64-channel block fits the 128 registers beside the next block's first weights.
        constexpr int P = 40;  // floats per output channel of a strip: the half waves (4 channels apart) land 32 banks apart
        float* mine = reinterpret_cast<float*>(region) + wave * (32 * P);
        const int q4 = lane & 7, rsub = lane >> 3;
        float bq[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int o = ocw + it * 8 + rsub;
            bq[it] = epi.bias ? epi.bias[o < g.oc ? o : g.oc - 1] : 0.0f;
        }
        int colq[NJ];
        bool every = true;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int p = (NJ * wn + j) * 32 + 4 * q4, row = wt_row(wt, p), oy = ty0 + row, oxq = tx0 + p - row * wt.tw;
            colq[j] = p < live && oy < g.oh && oxq < g.ow ? oy * g.ow + oxq : -1;
            every = every && colq[j] < (g.plane & ~7);  // four positions from a multiple of four: all on one side of the last multiple of eight
        }
        // the activation's scalar-tail form (libm) only where some lane is in the last 0-7 positions of the plane
        const bool all_body = __builtin_amdgcn_ballot_w64(!every) == 0;
        E_STAMP();  // e1: bias values requested, coordinates known
        float4 rv[NJ][4];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[((r & 3) + 8 * (r >> 2) + 4 * hv) * P + l31] = acc[j][r];
#pragma unroll
            for (int it = 0; it < 4; ++it)  // same-wave LDS order holds: no barrier, and the next strip's writes come after these reads
                rv[j][it] = *reinterpret_cast<const float4*>(mine + (it * 8 + rsub) * P + 4 * q4);
        }
        E_STAMP();  // e2: the strips' LDS turn is issued
        auto run = [&](auto fn) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                if (j == 1) E_STAMP();  // e3: strip 0 is out (waited for the bias values and the first pieces)
                const bool body = colq[j] < (g.plane & ~7);
                // a residual's four pieces of the strip are requested together, ahead of the activation: fetched where they are added, each
                // piece waited a round trip for its own load (vmcnt(0) sixteen times an item on the 18 bottleneck layers of the network)
                float4 r4[4];
                if (g.res) {  // uniform
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int oc = ocw + it * 8 + rsub;
                        const bool ok = colq[j] >= 0 && oc < g.oc;
                        r4[it] = *reinterpret_cast<const float4*>(g.res + (int64_t)img * g.rbs + (int64_t)(ok ? oc : 0) * g.plane + (ok ? colq[j] : 0));
                    }
                }
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int oc = ocw + it * 8 + rsub;
                    float4 v = rv[j][it];
                    if (epi.bias) v.x = v.x + bq[it], v.y = v.y + bq[it], v.z = v.z + bq[it], v.w = v.w + bq[it];
                    v.x = fn(v.x, body), v.y = fn(v.y, body), v.z = fn(v.z, body), v.w = fn(v.w, body);
                    if ((LELE_CONV_KO & 2) && v.x != 12345.678f) continue;
                    if (colq[j] >= 0 && oc < g.oc) {
                        LELE_DEV_ASSERT(colq[j] + 3 < g.plane && img >= 0 && img < g.n);
                        if (g.res) v.x = v.x + r4[it].x, v.y = v.y + r4[it].y, v.z = v.z + r4[it].z, v.w = v.w + r4[it].w;  // uniform
                        *reinterpret_cast<float4*>(epi.out + (int64_t)img * g.obs + (int64_t)oc * g.plane + colq[j]) = v;
                    }
                }
            }
        };
        if (epi.act == LELE_ACT_NONE) run([](float v, bool) { return v; });
        else if (epi.act == LELE_ACT_RELU) run([](float v, bool) { return v > 0.0f ? v : 0.0f; });
        else if (epi.act == LELE_ACT_SILU) run([](float v, bool) { return silu_fast(v); });
        else if (all_body) run([](float v, bool) { return apply_act(v, kActSiluExact, true); });
        else run([](float v, bool b) { return apply_act(v, kActSiluExact, b); });
        E_STAMP();  // e4: all stores issued
        return;
    }
    float bv[16];  // the scalar path: every value where it sits
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = ocw + (r & 3) + 8 * (r >> 2) + 4 * hv;
        bv[r] = epi.bias ? epi.bias[o < g.oc ? o : g.oc - 1] : 0.0f;
    }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int p = (NJ * wn + j) * 32 + l31, row = wt_row(wt, p), oy = ty0 + row, ox = tx0 + p - row * wt.tw;
        if (!(p < live && oy < g.oh && ox < g.ow)) continue;
        const int col = oy * g.ow + ox;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int oc = ocw + (r & 3) + 8 * (r >> 2) + 4 * hv;
            if (oc < g.oc) epi.store(img, oc, col, acc[j][r], bv[r]);
        }
    }
}

#undef E_STAMP
// Which items a persistent workgroup takes.  Workgroup b runs on XCD b % 8, and every XCD has its own L2: handed out round robin
// (b, b + G, ...), the tiles next to each other in an image -- whose windows share halo rows and, more to the point, the 128-byte lines
// their ragged row ends lie in -- are multiplied on eight different XCDs, and each fetches those lines again (counters on the
// reference graph at batch 64: the 3 x 3 layers read 2.4 x and the stride-2 layers 1.6-2.5 x their input from the fabric).  So every
// XCD gets one CONTIGUOUS share of the items and its workgroups walk it side by side: first = lo + (b / 8), step = the XCD's
// workgroup count.  (Grids of fewer than eight workgroups keep the plain order.)  Returns the number of items of this workgroup.
__device__ __forceinline__ int win_items(int items, int* first, int* step) {
    const int G = (int)gridDim.x, b = (int)blockIdx.x;
    if (G < 8) {
        *first = b;
        *step = G;
        return (items - b + G - 1) / G;
    }
    // XCD x holds gx = ceil((G - x) / 8) workgroups, cum of them sit on the XCDs before it; its range is that share of the items (the
    // grid never exceeds the item count, so every workgroup finds at least one)
    const int xcd = b & 7, j = b >> 3, gx = (G + 7 - xcd) >> 3, cum = xcd * (G >> 3) + (xcd < (G & 7) ? xcd : (G & 7));
    const int lo = (int)((int64_t)items * cum / G), hi = (int)((int64_t)items * (cum + gx) / G);
    *first = lo + j;
    *step = gx;
    return lo + j < hi ? (hi - (lo + j) + gx - 1) / gx : 0;
}

template <int KS, int OCT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_window_p_kernel(const float* __restrict__ x,
                                                                                                   const cu32x4* __restrict__ wfrag,
                                                                                                   ConvEpi epi, WinTile wt, int ntiles, int nocb,
                                                                                                   int osplit, int items) {
    typedef C3M<KS> W;
    // OCT = 128: the four multiplying waves take a block of 32 output channels each over a tile of 128 positions (the window is fetched
    // and split once for 128 output channels; 64- and 32-channel blocks of a 256-position tile fetch it once per block)
    constexpr int NJ = OCT == 32 ? 2 : 4, MTB = OCT / 32;
    static_assert(4 * 32 * 40 * 4 <= W::STAGE, "a strip of every consumer wave fits one stage");
    const ConvGeom& g = epi.g;
    extern __shared__ __attribute__((aligned(16))) char c3m_lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hv = lane >> 5, l31 = lane & 31;
    const int hw = g.ih * g.iw, nchunk = g.c / 16;
    const int pwt = wt.tw + KS - 1, post = (wt.th + KS - 1) * pwt;
    // An item = (image, tile, group of `nocb` consecutive blocks of output channels); `osplit` groups make up the layer's blocks
    // (1 when there are enough (image, tile) pairs to go round: the blocks then take turns inside the workgroup and the second one's
    // window comes out of L2; otherwise every block is an item of its own, so that a small layer still spreads over the chip)
    // this workgroup's items: first, first + G, ... below the end of its range (win_items: every XCD a contiguous range of items)
    int first, G;
    const int nitem = win_items(items, &first, &G);
    const int nseq = nitem * nocb;    // (item, block of output channels) pairs, in order
    const int qtotal = nseq * nchunk;  // chunks of the whole stream
    // (a raw s_barrier: the compiler's own would wait vmcnt(0), i.e. for every load in flight.  What it must wait for is this wave's LDS
    // traffic -- a parked chunk's ds_writes are only ISSUED when the instruction after them runs, and the barrier hands the stage over)
    auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
#ifdef LELE_HIP_LAB
    long long* const dbg_w = epi.dbg ? epi.dbg + ((size_t)blockIdx.x * 8 + wave) * 64 : nullptr;
    int dbg_n = 0;
#define C_STAMP()                                                              \
    do {                                                                       \
        if (dbg_w && dbg_n < 64 && lane == 0) dbg_w[dbg_n] = (long long)clock64(); \
        ++dbg_n;                                                               \
    } while (0)
#else
#define C_STAMP()
#endif
    C_STAMP();  // 0: entry
    if (wave >= 4) {
        // ------------------------------------------------------------ producers
        const int pt = tid - 256;
        int t_lds[W::TASKS];
        unsigned t_pq[W::TASKS], t_off[W::TASKS];  // (quad << 24 | row << 12 | column) of the task inside the window; byte offset for the item at hand
#pragma unroll
        for (int i = 0; i < W::TASKS; ++i) {
            const int t = pt + 256 * i, q = t / post, pos = t - q * post, py = pos / pwt, px = pos - py * pwt;
            t_pq[i] = q < 4 ? ((unsigned)q << 24) | ((unsigned)py << 12) | (unsigned)px : 0xffffffffu;
            t_lds[i] = q < 4 ? pos * C3M_PITCH + 8 * q : -1;
            LELE_DEV_ASSERT(post <= W::POS && py < 4096 && px < 4096 && t_lds[i] + 64 + 8 <= W::STAGE);
        }
        const float* xin = x;
        int f_item = first - G, f_ocb = nocb - 1, f_cc = nchunk - 1;  // the fetch cursor: one step before the first chunk
        auto fetch = [&](float4 (&st)[W::TASKS], bool live) {
            if (++f_cc == nchunk) {
                f_cc = 0;
                if (++f_ocb == nocb) {  // the next item: where its window lies
                    f_ocb = 0;
                    f_item += G;
                    const int pair = f_item / osplit;
                    const int img = pair / ntiles, tile = pair - img * ntiles, tyi = tile / wt.tiles_x, txi = tile - tyi * wt.tiles_x;
                    const int iy0 = tyi * wt.th - g.pt, ix0 = txi * wt.tw - g.pl;
                    xin = x + (int64_t)img * g.xbs;
#pragma unroll
                    for (int i = 0; i < W::TASKS; ++i) {
                        const int q = (int)(t_pq[i] >> 24), iy = iy0 + (int)((t_pq[i] >> 12) & 0xfffu), ix = ix0 + (int)(t_pq[i] & 0xfffu);
                        const bool in = t_pq[i] != 0xffffffffu && iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw;
                        t_off[i] = in ? (unsigned)(4 * q * hw + iy * g.iw + ix) * 4u : 0xfffffff0u;  // past the resource: reads as 0
                    }
                }
            }
            // one resource per channel of a quad (bases one plane apart, 13 planes long), as in the stride-2 kernel
            const float* cb = xin + (int64_t)f_cc * 16 * hw;
            // (`live` false: past the last chunk -- the same loads against an empty resource read zeros without touching memory.  The loads
            // must not sit under a branch: the compiler's in-order count of the loads in flight is exact only on straight-line code; with
            // `if (more) fetch()` every park waited for ALL D sets, vmcnt(0), i.e. for the chunk just requested)
            const int extent = (LELE_CONV_KO & 8) || !live ? 0 : (int)(13u * (unsigned)hw * 4u);
            const auto r0 = __builtin_amdgcn_make_buffer_rsrc((void*)cb, (short)0, extent, 0x00020000);
            const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(cb + hw), (short)0, extent, 0x00020000);
            const auto r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(cb + 2 * hw), (short)0, extent, 0x00020000);
            const auto r3 = __builtin_amdgcn_make_buffer_rsrc((void*)(cb + 3 * hw), (short)0, extent, 0x00020000);
#pragma unroll
            for (int i = 0; i < W::TASKS; ++i) {
                st[i].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, (int)t_off[i], 0, 0));
                st[i].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (int)t_off[i], 0, 0));
                st[i].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, (int)t_off[i], 0, 0));
                st[i].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r3, (int)t_off[i], 0, 0));
            }
        };
        auto park = [&](const float4 (&st)[W::TASKS], int buf) {
            char* dst = c3m_lds + buf * W::STAGE;
#pragma unroll
            for (int i = 0; i < W::TASKS; ++i) {
                if (t_lds[i] < 0) continue;
                unsigned h0, m0, l0, h1, m1, l1;  // three bf16 pieces a value, rounded to nearest (common.h split3_bf16_pair)
                split3_bf16_pair(st[i].x, st[i].y, h0, m0, l0);
                split3_bf16_pair(st[i].z, st[i].w, h1, m1, l1);
                const cu32x2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
                *reinterpret_cast<cu32x2*>(dst + t_lds[i]) = h;
                *reinterpret_cast<cu32x2*>(dst + t_lds[i] + 32) = m;
                *reinterpret_cast<cu32x2*>(dst + t_lds[i] + 64) = l;
            }
        };
        // D register sets: the loads of D chunks are in flight (in the network the windows come from HBM, not from the last-level cache
        // a repeated micro-benchmark reads them from: two chunks of 16 KB a workgroup did not cover the latency)
        constexpr int D = KS == 1 ? 4 : 3;
        float4 st[D][W::TASKS];
#pragma unroll
        for (int d = 0; d < D; ++d) fetch(st[d], d < qtotal);
        C_STAMP();  // 1: D chunks requested
        park(st[0], 0);
        C_STAMP();  // 2: chunk 0 parked
        barrier();  // B_0
        C_STAMP();  // 3
        // `since` = (q - 1) % nchunk for the chunk q about to be parked: 0 means chunk q - 2 ended an item's block, whose epilogue uses
        // the stage chunk q goes to -- wait for E first
        int since = 0;
        for (int q0 = 1; q0 < qtotal; q0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {  // chunk q0 + u sits in set (1 + u) % D; parking chunk q0 + u - 1 freed set u % D
                const int q = q0 + u;
                if (q >= qtotal) break;
                fetch(st[u % D], q - 1 + D < qtotal);
                C_STAMP();  // 4 + 3 (q - 1): requested
                if (q >= 2 && since == 0) barrier();  // E
                park(st[(1 + u) % D], q & 1);
                C_STAMP();  // 5 + 3 (q - 1): chunk q parked
                barrier();  // B_q
                C_STAMP();  // 6 + 3 (q - 1)
                if (++since == nchunk) since = 0;
            }
        }
        barrier();  // B_qtotal: the consumers' last "done"
        return;
    }
    // ---------------------------------------------------------------- consumers
    const int wm = OCT == 128 ? wave : OCT == 64 ? (wave & 1) : 0, wn = OCT == 128 ? 0 : OCT == 64 ? (wave >> 1) : wave;
    cf32x16 acc[NJ];
    int sb[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int p = (NJ * wn + j) * 32 + l31;
        p = p < wt.tw * wt.th ? p : 0;
        const int row = wt_row(wt, p);
        sb[j] = (row * pwt + p - row * wt.tw) * C3M_PITCH + hv * 16;
        LELE_DEV_ASSERT(sb[j] + ((KS - 1) * pwt + KS - 1) * C3M_PITCH + 64 + 16 <= W::STAGE);  // the last tap's last piece
    }
    cu32x4 ar[2][3];
    const int wtaps = nchunk * W::TAPS;  // weight fragments of one block of output channels
    auto wblock = [&](int blk) { return wfrag + ((int64_t)(blk * MTB + wm) * nchunk) * (W::TAPS * 3 * 64) + lane; };
    const cu32x4* wcur = wblock((first % osplit) * nocb);
    const cu32x4* wnxt = wcur;
    auto wload = [&](cu32x4 (&dst)[3], int gt) {  // fragment gt of the current block; one past its end: the next block's first
        const cu32x4* src = gt < wtaps ? wcur + (int64_t)gt * (3 * 64) : wnxt;
#ifdef LELE_HIP_LAB
        if (g.dh & 0x100) src = wcur;  // knock-out (LELE_HIP_CONV_KO=1): every fragment is the block's first -- what do the consumers' weight fetches cost?
#endif
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = src[p * 64];
    };
    int q = 0;  // chunks done
    auto chunk = [&](int cc, auto pc) {
        constexpr int P = decltype(pc)::value;
        const char* stage = c3m_lds + (q & 1) * W::STAGE;
#define LELE_CBF(v) __builtin_bit_cast(cbf16x8, v)
        constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};  // mm, hl, lh, hm, mh, hh: smallest terms first
        if constexpr (OCT == 32 && KS == 3) {
            // 32-channel blocks use 88 registers: room for a second set of B fragments, read a tap ahead of their products
            cu32x4 bfr[2][2][3];
            auto loadb = [&](cu32x4 (&bf)[2][3], int tap) {
                const int a = tap / KS, b = tap - KS * a;
                const char* tapw = stage + (a * pwt + b) * C3M_PITCH;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const cu32x4*>(tapw + sb[j] + 32 * p);
            };
            loadb(bfr[0], 0);
#pragma unroll
            for (int tap = 0; tap < W::TAPS; ++tap) {
                cu32x4 (&af)[3] = ar[(P + tap) & 1];
                wload(ar[(P + tap + 1) & 1], cc * W::TAPS + tap + 1);
                // the next tap's weights are REQUESTED here: left to itself the scheduler sinks the three loads behind the tap's products, and
                // the next tap then waits a whole round trip for them
                __builtin_amdgcn_sched_barrier(0);
                if (tap + 1 < W::TAPS) loadb(bfr[(tap + 1) & 1], tap + 1);
#pragma unroll
                for (int t = (LELE_CONV_KO & 4) ? 5 : 0; t < 6; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_CBF(af[PA[t]]), LELE_CBF(bfr[tap & 1][j][PB[t]]), acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
#pragma unroll
        for (int tap = 0; tap < W::TAPS; ++tap) {
            const int a = tap / KS, b = tap - KS * a;
            const char* tapw = stage + (a * pwt + b) * C3M_PITCH;
            cu32x4 (&af)[3] = ar[(P + tap) & 1];
            wload(ar[(P + tap + 1) & 1], cc * W::TAPS + tap + 1);
            // the next tap's weights are REQUESTED here: left to itself the scheduler sinks the three loads behind the tap's products, and
            // the next tap then waits a whole round trip for them
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int jp = 0; jp < NJ / 2; ++jp) {
                cu32x4 bf[2][3];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const char* src = tapw + sb[2 * jp + j];
#pragma unroll
                    for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const cu32x4*>(src + 32 * p);
                }
#pragma unroll
                for (int t = (LELE_CONV_KO & 4) ? 5 : 0; t < 6; ++t)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[2 * jp + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_CBF(af[PA[t]]), LELE_CBF(bf[j][PB[t]]), acc[2 * jp + j], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#undef LELE_CBF
    };
    int item = first, ocb = 0;
    wload(ar[0], 0);
    barrier();  // B_0
    C_STAMP();  // 1: chunk 0 is there
    for (int s = 0; s < nseq; ++s) {
        const int pair = item / osplit, grp = item - pair * osplit;
        const int img = pair / ntiles, tile = pair - img * ntiles, tyi = tile / wt.tiles_x, txi = tile - tyi * wt.tiles_x;
        const int blk = grp * nocb + ocb;  // this pass's block of output channels
        {
            const int item_n = ocb + 1 == nocb ? item + G : item;
            wnxt = wblock((item_n % osplit) * nocb + (ocb + 1 == nocb ? 0 : ocb + 1));  // (past the last item: any block, never multiplied)
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        // nine taps or one: the register sets alternate from chunk to chunk, so chunks go in pairs; a block with an odd count
        // leaves the next block's first fragment in set 1 and moves it (every block starts from set 0)
        int cc = 0;
        for (; cc + 1 < nchunk; cc += 2) {
            chunk(cc, std::integral_constant<int, 0>());
            C_STAMP();  // products issued
            barrier();  // B_{q + 1}: done with this stage, and the next chunk is in the other one
            C_STAMP();
            ++q;
            chunk(cc + 1, std::integral_constant<int, 1>());
            C_STAMP();
            barrier();
            C_STAMP();
            ++q;
        }
        if (cc < nchunk) {
            chunk(cc, std::integral_constant<int, 0>());
            C_STAMP();
            barrier();
            C_STAMP();
            ++q;
#pragma unroll
            for (int p = 0; p < 3; ++p) ar[0][p] = ar[1][p];
        }
#ifdef LELE_HIP_LAB
        c3m_epilogue_strips<NJ, OCT>(acc, epi, g, c3m_lds + ((q - 1) & 1) * W::STAGE, wave, lane, wm, wn, blk, img, tyi * wt.th, txi * wt.tw, wt, dbg_w, &dbg_n);
#else
        c3m_epilogue_strips<NJ, OCT>(acc, epi, g, c3m_lds + ((q - 1) & 1) * W::STAGE, wave, lane, wm, wn, blk, img, tyi * wt.th, txi * wt.tw, wt);
#endif
        C_STAMP();  // strips out
        if (q + 1 < qtotal) barrier();  // E: chunk q + 1 may now be parked where the strips went
        C_STAMP();
        wcur = wnxt;
        if (++ocb == nocb) ocb = 0, item += G;
    }
}
#undef C_STAMP
// ---- the same for STRIDE 2 (3 x 3): a workgroup owns 64 output channels x (4 rows x 32 columns) of one image.  The window of a
// stride-2 tile is 9 x 65 input positions; stored as it lies, tap (a, b) of output column l would read position 2 l + b -- a
// 224-byte lane stride, two lanes per bank.  The producers therefore store the window DE-INTERLEAVED into its four phases
// (row parity, column parity): position (py, px) goes to plane 2 (py & 1) + (px & 1), slot (py >> 1, px >> 1), and tap (a, b) of
// output (j, l) reads plane 2 (a & 1) + (b & 1) at slot (j + (a >> 1), l + (b >> 1)) -- consecutive lanes, consecutive slots, the
// conflict-free 112-byte pitch of the stride-1 kernel.  660 slots x 112 bytes = 74 KB: ONE stage per workgroup and two workgroups
// per CU (the stride-1 kernel's measurement: occupancy, not a second stage, is what hides a workgroup's staging and epilogue) --
// the producers keep the next chunk in registers, park it when the consumers are done with the stage, and the other workgroup of
// the CU multiplies meanwhile.  Same weight fragments and six-term products as the stride-1 kernel.
struct C3S2 {
    static constexpr int TH = 4, TW = 32, PH = 2 * (TH - 1) + 3, PW = 2 * (TW - 1) + 3;   // 9 x 65 input positions
    static constexpr int SY = (PH + 1) / 2, SX = (PW + 1) / 2, PLANE = SY * SX, POS = 4 * PLANE;   // 5 x 33 slots per phase plane
    static constexpr int STAGE = POS * C3M_PITCH;
    static constexpr int TASKS = (POS * 4 + 255) / 256;
};
// OCT = 64: two 32-channel tiles x two pairs of strips; OCT = 32: one tile x four single strips (narrow layers).
// PERSISTENT like the stride-1 kernel (same items, same schedule): one stream of chunks across items, the loaders a chunk ahead in
// registers -- the next item's first window is in flight during the epilogue -- and the strips leave through the (single) stage once
// every multiplier is done with the item's last chunk.  Barriers per chunk: "parked" and "done"; per item two more around the epilogue.
template <int OCT>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void conv_window_s2_kernel(const float* __restrict__ x,
                                                                                                    const cu32x4* __restrict__ wfrag,
                                                                                                    ConvEpi epi, WinTile wt, int ntiles, int nocb,
                                                                                                    int osplit, int items) {
    typedef C3S2 W;
    constexpr int NJ = OCT == 64 ? 2 : 1, MTB = OCT / 32, TAPS = 9;
    static_assert(4 * 32 * 40 * 4 <= W::STAGE, "a strip of every consumer wave fits the stage");
    extern __shared__ __attribute__((aligned(16))) char c3m_lds[];
    const ConvGeom& g = epi.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hv = lane >> 5, l31 = lane & 31;
    const int hw = g.ih * g.iw, nchunk = g.c / 16;
    const int sxt = wt.tw + 1, planet = (wt.th + 1) * sxt, post = 4 * planet;  // slots of a phase plane for this tile shape: post <= W::POS
    int first, G;
    const int nitem = win_items(items, &first, &G);
    const int nseq = nitem * nocb, qtotal = nseq * nchunk;
    // (a raw s_barrier: the compiler's own would wait vmcnt(0), i.e. for every load in flight.  What it must wait for is this wave's LDS
    // traffic -- a parked chunk's ds_writes are only ISSUED when the instruction after them runs, and the barrier hands the stage over)
    auto barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
    if (wave >= 4) {
        // ------------------------------------------------------------ producers: one chunk in registers, parked when the stage is free
        const int pt = tid - 256;
        int t_lds[W::TASKS];
        unsigned t_pq[W::TASKS], t_off[W::TASKS];
#pragma unroll
        for (int i = 0; i < W::TASKS; ++i) {
            const int t = pt + 256 * i, q = t / post, slot = t - q * post;  // q < 4 while t < 4 * post
            const int ph = slot / planet, r = slot - ph * planet, sy = r / sxt, sx = r - sy * sxt;
            const int py = 2 * sy + (ph >> 1), px = 2 * sx + (ph & 1);
            const bool any = q < 4 && py < 2 * wt.th + 1 && px < 2 * wt.tw + 1;  // (the last row / column of the odd planes does not exist)
            t_pq[i] = any ? ((unsigned)q << 24) | ((unsigned)py << 12) | (unsigned)px : 0xffffffffu;
            t_lds[i] = q < 4 ? slot * C3M_PITCH + 8 * q : -1;
        }
        const float* xin = x;
        int f_item = first - G, f_ocb = nocb - 1, f_cc = nchunk - 1;
        float4 st[W::TASKS];
        auto fetch = [&] {
            if (++f_cc == nchunk) {
                f_cc = 0;
                if (++f_ocb == nocb) {
                    f_ocb = 0;
                    f_item += G;
                    const int pair = f_item / osplit;
                    const int img = pair / ntiles, tile = pair - img * ntiles, tyi = tile / wt.tiles_x, txi = tile - tyi * wt.tiles_x;
                    const int iy0 = tyi * wt.th * 2 - g.pt, ix0 = txi * wt.tw * 2 - g.pl;
                    xin = x + (int64_t)img * g.xbs;
#pragma unroll
                    for (int i = 0; i < W::TASKS; ++i) {
                        const int q = (int)(t_pq[i] >> 24), iy = iy0 + (int)((t_pq[i] >> 12) & 0xfffu), ix = ix0 + (int)(t_pq[i] & 0xfffu);
                        const bool in = t_pq[i] != 0xffffffffu && iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw;
                        t_off[i] = in ? (unsigned)(4 * q * hw + iy * g.iw + ix) * 4u : 0xfffffff0u;  // past the resource: reads as 0
                    }
                }
            }
            const float* cb = xin + (int64_t)f_cc * 16 * hw;
            const int extent = (int)(13u * (unsigned)hw * 4u);
            const auto r0 = __builtin_amdgcn_make_buffer_rsrc((void*)cb, (short)0, extent, 0x00020000);
            const auto r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(cb + hw), (short)0, extent, 0x00020000);
            const auto r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(cb + 2 * hw), (short)0, extent, 0x00020000);
            const auto r3 = __builtin_amdgcn_make_buffer_rsrc((void*)(cb + 3 * hw), (short)0, extent, 0x00020000);
#pragma unroll
            for (int i = 0; i < W::TASKS; ++i) {
                st[i].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r0, (int)t_off[i], 0, 0));
                st[i].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r1, (int)t_off[i], 0, 0));
                st[i].z = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r2, (int)t_off[i], 0, 0));
                st[i].w = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r3, (int)t_off[i], 0, 0));
            }
        };
        auto park = [&] {
#pragma unroll
            for (int i = 0; i < W::TASKS; ++i) {
                if (t_lds[i] < 0) continue;
                unsigned h0, m0, l0, h1, m1, l1;  // three bf16 pieces a value, rounded to nearest (common.h split3_bf16_pair)
                split3_bf16_pair(st[i].x, st[i].y, h0, m0, l0);
                split3_bf16_pair(st[i].z, st[i].w, h1, m1, l1);
                const cu32x2 h = {h0, h1}, m = {m0, m1}, l = {l0, l1};
                *reinterpret_cast<cu32x2*>(c3m_lds + t_lds[i]) = h;
                *reinterpret_cast<cu32x2*>(c3m_lds + t_lds[i] + 32) = m;
                *reinterpret_cast<cu32x2*>(c3m_lds + t_lds[i] + 64) = l;
            }
        };
        fetch();
        int since = 0;  // chunks of the current block parked so far
        for (int q = 0; q < qtotal; ++q) {
            park();
            barrier();  // chunk q is parked
            if (q + 1 < qtotal) fetch();
            barrier();  // the multipliers are done with chunk q
            if (++since == nchunk) {
                since = 0;
                barrier();  // ... and with the block's epilogue, which went through the stage
            }
        }
        return;
    }
    // ---------------------------------------------------------------- consumers: 32 output channels x NJ strips of the tile each
    const int wm = OCT == 64 ? (wave & 1) : 0, wn = OCT == 64 ? (wave >> 1) : wave;
    cf32x16 acc[NJ];
    int sb[NJ];  // this lane's position of strip j: byte offset of its slot (row, column) in a phase plane
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int p = (NJ * wn + j) * 32 + l31;
        p = p < wt.tw * wt.th ? p : 0;
        const int row = wt_row(wt, p);
        sb[j] = (row * sxt + p - row * wt.tw) * C3M_PITCH + hv * 16;
    }
    cu32x4 ar[2][3];
    const int wtaps = nchunk * TAPS;
    auto wblock = [&](int blk) { return wfrag + ((int64_t)(blk * MTB + wm) * nchunk) * (TAPS * 3 * 64) + lane; };
    const cu32x4* wcur = wblock((first % osplit) * nocb);
    const cu32x4* wnxt = wcur;
    auto wload = [&](cu32x4 (&dst)[3], int gt) {
        const cu32x4* src = gt < wtaps ? wcur + (int64_t)gt * (3 * 64) : wnxt;
#pragma unroll
        for (int p = 0; p < 3; ++p) dst[p] = src[p * 64];
    };
    auto chunk = [&](int cc, auto pc) {
        constexpr int P = decltype(pc)::value;
        barrier();  // the chunk is parked
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int a = tap / 3, b = tap - 3 * a;
            const char* tapw = c3m_lds + ((2 * (a & 1) + (b & 1)) * planet + (a >> 1) * sxt + (b >> 1)) * C3M_PITCH;
            cu32x4 (&af)[3] = ar[(P + tap) & 1];
            wload(ar[(P + tap + 1) & 1], cc * TAPS + tap + 1);
            // the next tap's weights are REQUESTED here: left to itself the scheduler sinks the three loads behind the tap's products, and
            // the next tap then waits a whole round trip for them
            __builtin_amdgcn_sched_barrier(0);
#define LELE_CBF(v) __builtin_bit_cast(cbf16x8, v)
            constexpr int PA[6] = {1, 0, 2, 0, 1, 0}, PB[6] = {1, 2, 0, 1, 0, 0};  // mm, hl, lh, hm, mh, hh: smallest terms first
            cu32x4 bf[NJ][3];
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const char* src = tapw + sb[j];
#pragma unroll
                for (int p = 0; p < 3; ++p) bf[j][p] = *reinterpret_cast<const cu32x4*>(src + 32 * p);
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(LELE_CBF(af[PA[t]]), LELE_CBF(bf[j][PB[t]]), acc[j], 0, 0, 0);
#undef LELE_CBF
            __builtin_amdgcn_sched_barrier(0);
        }
        barrier();  // done with the chunk
    };
    int item = first, ocb = 0;
    wload(ar[0], 0);
    for (int s = 0; s < nseq; ++s) {
        const int pair = item / osplit, grp = item - pair * osplit;
        const int img = pair / ntiles, tile = pair - img * ntiles, tyi = tile / wt.tiles_x, txi = tile - tyi * wt.tiles_x;
        const int blk = grp * nocb + ocb;
        {
            const int item_n = ocb + 1 == nocb ? item + G : item;
            wnxt = wblock((item_n % osplit) * nocb + (ocb + 1 == nocb ? 0 : ocb + 1));
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
        int cc = 0;
        for (; cc + 1 < nchunk; cc += 2) {
            chunk(cc, std::integral_constant<int, 0>());
            chunk(cc + 1, std::integral_constant<int, 1>());
        }
        if (cc < nchunk) {
            chunk(cc, std::integral_constant<int, 0>());
#pragma unroll
            for (int p = 0; p < 3; ++p) ar[0][p] = ar[1][p];
        }
        c3m_epilogue_strips<NJ, OCT>(acc, epi, g, c3m_lds, wave, lane, wm, wn, blk, img, tyi * wt.th, txi * wt.tw, wt);
        barrier();  // the strips are out: the stage is the loaders' again
        wcur = wnxt;
        if (++ocb == nocb) ocb = 0, item += G;
    }
}
// weights [OC][IC][taps] f32 -> split-bf16 fragments [ceil(OC / 32)][IC / 16][taps][3 pieces][64 lanes] x 16 bytes (zeros for the
// channels past OC): lane (l31 = output channel in the tile, hv) holds input channels 16 chunk + 8 hv + [0, 8) of its tap
__global__ void conv_wfrag_kernel(const float* __restrict__ w, cu32x4* __restrict__ wfrag, int oc, int ic, int taps) {
    const int64_t total = (int64_t)((oc + 31) / 32) * (ic / 16) * taps * 64;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int lane = (int)(i & 63), tap = (int)((i >> 6) % taps);
        const int64_t rest = (i >> 6) / taps;
        const int cc = (int)(rest % (ic / 16)), mt = (int)(rest / (ic / 16));
        const int o = mt * 32 + (lane & 31), c0 = cc * 16 + 8 * (lane >> 5);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = o < oc ? w[((int64_t)o * ic + c0 + e) * taps + tap] : 0.0f;
        cu32x4 h, m, l;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned hp, mp, lp;
            split3_bf16_pair(v[2 * p], v[2 * p + 1], hp, mp, lp);
            h[p] = hp, m[p] = mp, l[p] = lp;
        }
        cu32x4* dst = wfrag + (((int64_t)mt * (ic / 16) + cc) * taps + tap) * (3 * 64) + lane;
        dst[0] = h;
        dst[64] = m;
        dst[128] = l;
    }
}

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192)); }
inline int64_t attr(const int64_t* v, size_t n, size_t i, int64_t dflt) {
    if (n >= 2) return v[i];  // conv2d.rs:208-243: two values, or one value used for both
    if (n == 1) return v[0];
    return dflt;
}

// where the window-once kernel takes a 1 x 1 convolution (see the dispatch below); LELE_HIP_CONV_W1_* override them in the developer's build
inline int conv_env(const char* name, int dflt) {
    const char* v = lab_env(name);  // the developer's build only (common.h): the product library has the defaults compiled in
    return v && *v ? atoi(v) : dflt;
}
// (measured on the Yolo-shaped network at batch 64, whole forward.  One workgroup per tile, round 4's first half: <= 64 channels /
// planes >= 6400 / >= 48 input channels 10.12 ms; <= 128 / >= 1600 / >= 32: 9.93; <= 256 / >= 1600: 9.95; <= 256 / >= 400: 9.97 -- every
// further block of output channels fetched and split the window again.  Persistent workgroups, where the blocks of an item take turns
// on a window that is in L2: <= 128 / >= 1600 8.69; <= 256 / >= 1600 8.66; <= 128 / >= 400 8.58; <= 256 / >= 400 **8.25-8.38**;
// <= 512 / >= 400 8.28; <= 256 / >= 100 8.27; >= 16 input channels 8.27)
inline int w1_max_oc() { static const int v = conv_env("LELE_HIP_CONV_W1_MAXOC", 256); return v; }
inline int w1_min_plane() { static const int v = conv_env("LELE_HIP_CONV_W1_MINPLANE", 400); return v; }
// <= 16 output channels: the direct kernel multiplies on the vector pipe (C x 9 x OC FMAs a pixel), so from 32 input channels on a
// half-empty MFMA tile is faster for 9-16 output channels (at batch 64: 32 -> 16 at 80 x 80 85 -> 70 us, 64 -> 16 at 40 x 40 79 -> 32;
// 32 -> 8 at 80 x 80 stays direct: 62 against 67).  Round 6, after the window kernels' loaders and epilogue changed: from 16 input channels on
// (the 16 -> 16 layers of the reference graph): the graph at batch 64 7.87 -> 7.74 ms linear, three interleaved pairs
// and from 5 output channels on (win_min_oc_narrow: 32 -> 8 at 80 x 80 69 -> 51 us, 16 -> 8 at 160 x 160 104 -> 98)
inline int win_min_oc_narrow() { static const int v = conv_env("LELE_HIP_CONV_WIN_NARROW_MINOC", 4); return v; }
inline int win_min_c_narrow() { static const int v = conv_env("LELE_HIP_CONV_WIN_NARROW_MINC", 16); return v; }
inline int w1_min_c() { static const int v = conv_env("LELE_HIP_CONV_W1_MINC", 32); return v; }

// out += res, image by image: the residual of lele_hip_conv2d_res behind the depthwise kernels (which have their own epilogues);
// everything else adds it where the finished values are stored
__global__ __launch_bounds__(256) void conv_residual_kernel(float* __restrict__ out, const float* __restrict__ res, unsigned per_image,
                                                            long long obs, long long rbs) {
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < per_image; i += gridDim.x * 256u)
        out[(long long)blockIdx.y * obs + i] = out[(long long)blockIdx.y * obs + i] + res[(long long)blockIdx.y * rbs + i];
}

// The tile shape of a window-once launch (see WinTile): the fewest workgroups that cover an ow x oh map with th x tw <= `positions`
// output positions each, whose window fits `max_slots` LDS slots -- (th + 2)(tw + 2) at stride 1, four phase planes of (th + 1)(tw + 1)
// at stride 2, th x tw for a 1 x 1.  Widths are multiples of four (16-byte stores); among equal counts the widest tile (longest
// contiguous runs on both sides).  In the developer's build LELE_HIP_CONV_TILE=tw,th (or `rows`: round 3's fixed tiles) overrides it.
inline int narrow_beta() { static const int v = conv_env("LELE_HIP_CONV_TILE_BETA", 4); return v; }
inline WinTile pick_win_tile(int ow, int oh, int positions, int ks, int stride, int max_slots, int64_t units, int num_cus) {
    // `units` = images x blocks of output channels: units x tiles workgroups are launched.  A workgroup multiplies its `positions`
    // whether they are outputs or padding, so with more workgroups than CUs the fewest tiles win; a launch that does not even give
    // every CU one workgroup takes as long as ONE of them, and then the smallest window (the most tiles that still fit) is fastest
    // (256 -> 64 channels at 20 x 20 x 64 images: two 20 x 12 tiles an image 122 us, three 8-row tiles 105)
    auto slots = [&](int tw, int th) { return stride == 2 ? 4 * (th + 1) * (tw + 1) : (th + ks - 1) * (tw + ks - 1); };
    int btw = 0, bth = 0;
    int64_t best = -1, best_slots = 0;
    // stride 1 only: at stride 2 the four phase planes already make the rows short, and the same weight costs 64 -> 64 at 80 x 80 10 %
    const int beta = stride == 1 ? narrow_beta() : 0;
    for (int tw = 4; tw <= positions; tw += 4) {
        int th = std::min(positions / tw, oh);
        while (th >= 1 && slots(tw, th) > max_slots) --th;
        for (; th >= 1; --th) {
            const int64_t count = (int64_t)((ow + tw - 1) / tw) * ((oh + th - 1) / th);
            // ... weighted by 1 + 4 / tw: the window's rows are the contiguous runs of the loaders' and the epilogue's memory accesses
            // (80-wide maps at batch 64: 28 tiles of 40 x 6 against 25 of 16 x 16: -2 ... -13 % on every 3 x 3 layer)
            const int64_t cost = count * units <= num_cus ? (int64_t)num_cus * 64 : count * units * (tw + beta) * 64 / tw, sl = slots(tw, th);
            if (best < 0 || cost < best || (cost == best && (count * units <= num_cus ? sl <= best_slots : true)))
                best = cost, best_slots = sl, btw = tw, bth = th;
            if (count * units > num_cus) break;  // shorter tiles only add workgroups from here
        }
        if (tw >= ow) break;
    }
    static const char* forced = lab_env("LELE_HIP_CONV_TILE");
    if (forced && !strcmp(forced, "rows") && ks == 3) {  // the fixed tiles of round 3: 8 (4 at stride 2) rows of 32 columns
        btw = 32, bth = stride == 2 ? 4 : 8;
    } else if (forced && *forced && ks == 3) {
        int ftw = 0, fth = 0;
        if (sscanf(forced, "%d,%d", &ftw, &fth) == 2 && ftw >= 4 && ftw % 4 == 0 && fth >= 1 && ftw * fth <= positions && slots(ftw, fth) <= max_slots)
            btw = ftw, bth = fth;
    }
    return WinTile{btw, bth, (ow + btw - 1) / btw, (unsigned)(((1u << 20) + btw - 1) / btw)};
}

int run_conv2d(LeleCtx* ctx, const LeleTensor* wt, const float* dx, const float* dw, const float* db, ConvGeom g, int act,
               float* out) {
    if ((int64_t)g.n * g.oc * g.plane == 0) return 0;
    const bool pitched = (g.xbs != 0 && g.xbs != (long long)g.c * g.ih * g.iw) || (g.obs != 0 && g.obs != (long long)g.oc * g.plane);
    if (g.xbs == 0) g.xbs = (long long)g.c * g.ih * g.iw;
    if (g.obs == 0) g.obs = (long long)g.oc * g.plane;
    LELE_REQUIRE(!pitched || g.group == 1, "conv2d: channel views (a batch pitch) are supported for group == 1 only");
    if (g.icg == 1 && g.ocg == 1 && !pitched) {
        const int64_t total = (int64_t)g.n * g.oc * g.plane;
        LELE_REQUIRE(total < (int64_t(1) << 32), "depthwise conv: more than 2^32 output elements");
        const unsigned ngroups = (unsigned)((g.ow + 3) / 4);
        const int64_t threads = (int64_t)g.n * g.oc * g.oh * ngroups;
        const DwGeom dg{g.ih, g.iw, g.oh, g.ow, g.oc, g.kh, g.sh, g.dh, g.pt, g.pl, g.plane & ~7};
        const dim3 rgrid((unsigned)((threads + 255) / 256));
#define LELE_DW_ROW(KW) \
    hipLaunchKernelGGL(depthwise_row4_kernel<KW>, rgrid, dim3(256), 0, ctx->stream, dx, dw, db, out, dg, act, (unsigned)threads, ngroups)
        const bool row_ok = g.sw == 1 && g.dw == 1 && g.ow >= 8 && threads < (int64_t(1) << 31);
        // planes per workgroup for the LDS-staged form: input + output planes within 64 KB, enough workgroups to fill the chip
        // (rows a multiple of four wide are stored straight from registers: then only the input planes need LDS)
        const bool dw_direct = (g.ow & 3) == 0 && (g.plane & 3) == 0 && (((uintptr_t)out) & 15) == 0;
        const int64_t ihw = (int64_t)g.ih * g.iw, plane_floats = ihw + (dw_direct ? 0 : g.plane) + 8, planes = (int64_t)g.n * g.oc;
        int64_t pb = (16 * 1024 - 8) / plane_floats;
        // ... but no more planes than 16 KB worth (one plane if it is larger): a workgroup copies its planes in and only then computes, so
        // what overlaps one workgroup's copy with another's arithmetic is the number of workgroups a CU holds (80 x 80 planes over 64
        // images: two planes a workgroup 119 us, one 84; 40 x 40: five 44 us, two 37.5)
        if (pb >= 1) pb = std::max<int64_t>(1, std::min<int64_t>(pb, ((int64_t)conv_env("LELE_HIP_DW_LDS_KB", 16) * 256 - 8) / plane_floats));
        while (pb > 1 && (planes + pb - 1) / pb < 4 * (int64_t)ctx->num_cus) pb = (pb + 1) / 2;
        const bool lds_ok = row_ok && pb >= 1 && (g.kw == 3 || g.kw == 5 || g.kw == 7 || g.kw == 11) && !lab_env("LELE_HIP_DW_NO_LDS");
        if (lds_ok) {
            const size_t lds = (size_t)((((pb * ihw + 3) & ~int64_t(3)) + (dw_direct ? 0 : pb * g.plane)) * 4);
            const dim3 lgrid((unsigned)((planes + pb - 1) / pb));
#define LELE_DW_LDS(KW) \
    hipLaunchKernelGGL(depthwise_lds_kernel<KW>, lgrid, dim3(256), lds, ctx->stream, dx, dw, db, out, dg, act, (unsigned)planes, \
                       (unsigned)pb, ngroups)
            if (g.kw == 3) LELE_DW_LDS(3);
            else if (g.kw == 5) LELE_DW_LDS(5);
            else if (g.kw == 7) LELE_DW_LDS(7);
            else LELE_DW_LDS(11);
#undef LELE_DW_LDS
        } else if (row_ok && g.kw == 3) {
            LELE_DW_ROW(3);
        } else if (row_ok && g.kw == 5) {
            LELE_DW_ROW(5);
        } else if (row_ok && g.kw == 7) {
            LELE_DW_ROW(7);
        } else if (row_ok && g.kw == 11) {
            LELE_DW_ROW(11);
#undef LELE_DW_ROW
        } else {
            hipLaunchKernelGGL(depthwise_conv2d_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, dx, dw, db, out, g, act,
                               (unsigned)total);
        }
        if (g.res)
            hipLaunchKernelGGL(conv_residual_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)g.oc * g.plane + 255) / 256, 1024), (unsigned)g.n),
                               dim3(256), 0, ctx->stream, out, g.res, (unsigned)((int64_t)g.oc * g.plane), g.obs, g.rbs);
    } else if (g.group == 1 &&
               ((g.kh == 3 && g.kw == 3) ||
                // 1 x 1: where it measured faster than the tiled GEMM on the Yolo-shaped network at batch 64 (w1_max_oc() and friends above)
                (g.kh == 1 && g.kw == 1 && g.pt == 0 && g.pl == 0 && g.oh == g.ih && g.ow == g.iw && g.oc <= w1_max_oc() && g.plane >= w1_min_plane() &&
                 g.c >= w1_min_c())) &&
               g.dh == 1 && g.dw == 1 && g.sh == 1 && g.sw == 1 && g.c % 16 == 0 && (g.oc > 16 || (g.oc > win_min_oc_narrow() && g.c >= win_min_c_narrow())) && g.ow >= 16 && g.n <= 65535 &&
               (int64_t)g.c * g.ih * g.iw < (int64_t(1) << 31) &&
               (int64_t)g.n * ((g.oc + 63) / 64) * (((int64_t)g.plane + 255) / 256) >= (int64_t)ctx->num_cus / 2) {
        // stride 1 over a batch, 16-channel chunks: the window-once MFMA kernel (see conv_window_p_kernel); 32-channel blocks when that
        // wastes fewer output channels than 64-channel ones
        const int taps = g.kh * g.kw;
        // ... and when 64-channel blocks would not give every CU its two workgroups (256 -> 64 channels at 20 x 20 x 64 images: 115 us
        // with 256 workgroups of 64 channels, 78 with 512 of 32)
        int oct = ((g.oc + 31) / 32) * 32 < ((g.oc + 63) / 64) * 64 ? 32 : 64;
        if (oct == 64) {
            const int ow_ = taps == 1 ? g.plane : g.ow, oh_ = taps == 1 ? 1 : g.oh;
            const WinTile t64 = pick_win_tile(ow_, oh_, 256, g.kh, 1, taps == 9 ? C3M<3>::POS : C3M<1>::POS, (int64_t)g.n * ((g.oc + 63) / 64), ctx->num_cus);
            if ((int64_t)g.n * ((g.oc + 63) / 64) * t64.tiles_x * ((oh_ + t64.th - 1) / t64.th) < 2 * (int64_t)ctx->num_cus) oct = 32;
        }
        // 1 x 1 with more than one block of output channels: blocks of 128 over tiles of 128 positions, so that the window is fetched and
        // split once per 128 channels (where the tiles still go round the chip twice, and no more than a quarter of the block is padding).
        // Same call, batch 64: 96 -> 128 at 80 x 80 148 -> 139 us, 384 -> 128 at 40 x 40 92 -> 82, 512 / 384 / 256 / 128 -> 256 at 20 x 20
        // -10 ... -11 %; 80 output channels (48 of 128 padding) +4 ... +7 %: those keep their 64- / 32-channel blocks
        if (taps == 1 && (g.oc + oct - 1) / oct >= 2 && ((g.oc + 127) / 128) * 128 * 4 <= g.oc * 5 && conv_env("LELE_HIP_CONV_OCT128", 1) != 0 &&
            (int64_t)g.n * ((g.oc + 127) / 128) * (((int64_t)g.plane + 127) / 128) >= 2 * (int64_t)ctx->num_cus)
            oct = 128;
        const int positions = oct == 128 ? 128 : 256;
        const size_t wbytes = oct == 128 ? (size_t)(((g.oc + 127) / 128) * 4) * (g.c / 16) * taps * 3 * 1024  // whole blocks: the padding tiles are zeros
                                         : (size_t)((g.oc + 31) / 32) * (g.c / 16) * taps * 3 * 1024 + (size_t)(g.c / 16) * taps * 3 * 1024;  // (+ one padding tile)
        void* dwf = nullptr;
        const bool cacheable = wt->mem == LELE_MEM_WEIGHT;
        auto key = std::make_tuple((const void*)wt->data, wbytes, 330 + taps);
        auto it = cacheable ? ctx->weights.find(key) : ctx->weights.end();
        if (it != ctx->weights.end()) {
            dwf = it->second;
        } else {
            if (cacheable) {
                LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
                LELE_HIP_CHECK(hipMalloc(&dwf, wbytes));
                ctx->weights[key] = dwf;
            } else {
                LELE_TRY(ctx->arena_alloc(wbytes, &dwf));
            }
            // fragments for ceil(OC / 32) tiles, rounded up to whole blocks (the padding tiles are zeros)
            const int oc_pad = ((g.oc + oct - 1) / oct) * oct;
            LELE_HIP_CHECK(hipMemsetAsync(dwf, 0, wbytes, ctx->stream));
            hipLaunchKernelGGL(conv_wfrag_kernel, dim3(grid_for((int64_t)(oc_pad / 32) * (g.c / 16) * taps * 64)), dim3(256), 0, ctx->stream, dw,
                               (cu32x4*)dwf, g.oc, g.c, taps);
        }
        // a 1 x 1 convolution has no rows: its plane is ONE row, cut into tiles of 256 positions (only the last one has padding)
        if (taps == 1) {
            g.ih = g.oh = 1;
            g.iw = g.ow = g.plane;
        }
#ifdef LELE_HIP_LAB
        g.dh |= conv_env("LELE_HIP_CONV_KO", 0) << 8;  // the window kernels never read the dilation (it is 1 here): lab knock-out flags ride in it
#endif
        ConvEpi epi{out, db, g, act};
#ifdef LELE_HIP_LAB
        if (const char* e = lab_env("LELE_HIP_CONV_STAMPS")) epi.dbg = (long long*)(uintptr_t)strtoull(e, nullptr, 0);
#endif
        const WinTile tile = pick_win_tile(g.ow, g.oh, positions, g.kh, 1, taps == 9 ? C3M<3>::POS : C3M<1>::POS, (int64_t)g.n * ((g.oc + oct - 1) / oct),
                                           ctx->num_cus);
        // at most two workgroups per CU, each walking through its share of the items (see conv_window_p_kernel)
        const int ntiles = tile.tiles_x * ((g.oh + tile.th - 1) / tile.th), nblocks = (g.oc + oct - 1) / oct;
        const int osplit = (int64_t)g.n * ntiles >= 2 * (int64_t)ctx->num_cus ? 1 : nblocks, nocb = nblocks / osplit;
        const int64_t items = (int64_t)g.n * ntiles * osplit;
        LELE_REQUIRE(items < (int64_t(1) << 31), "conv2d: more than 2^31 tiles");
        const dim3 pgrid((unsigned)std::min<int64_t>(items, conv_env("LELE_HIP_CONV_WG_PER_CU", 2) * (int64_t)ctx->num_cus));
#define LELE_CW(KS_, OCT_)                                                                                                              \
    do {                                                                                                                                \
        auto kern = conv_window_p_kernel<KS_, OCT_>;                                                                                     \
        LELE_HIP_CHECK(lele::ensure_dyn_lds(reinterpret_cast<const void*>(kern), 2 * C3M<KS_>::STAGE));                                  \
        hipLaunchKernelGGL(kern, pgrid, dim3(512), 2 * C3M<KS_>::STAGE, ctx->stream, dx, (const cu32x4*)dwf, epi, tile, ntiles, nocb, osplit, \
                           (int)items);                                                                                                 \
    } while (0)
        if (taps == 9) {
            if (oct == 64) LELE_CW(3, 64);
            else LELE_CW(3, 32);
        } else {
            if (oct == 128) LELE_CW(1, 128);
            else if (oct == 64) LELE_CW(1, 64);
            else LELE_CW(1, 32);
        }
#undef LELE_CW
    } else if (g.group == 1 && g.kh == 3 && g.kw == 3 && g.dh == 1 && g.dw == 1 && g.sh == 2 && g.sw == 2 && g.c % 16 == 0 && g.oc > 16 &&
               g.ow >= 16 && g.n <= 65535 && (int64_t)g.c * g.ih * g.iw < (int64_t(1) << 31) &&
               (int64_t)g.n * ((g.oc + 63) / 64) * (((int64_t)g.plane + 127) / 128) >= 2 * (int64_t)ctx->num_cus &&
               !lab_env("LELE_HIP_CONV_NO_S2_WINDOW")) {
        // stride 2 over a batch: the de-interleaved window kernel (see conv_window_s2_kernel); weights as for the stride-1 kernel,
        // blocks of 32 output channels when that wastes fewer of them than blocks of 64
        const int oct = ((g.oc + 31) / 32) * 32 < ((g.oc + 63) / 64) * 64 ? 32 : 64;
        const size_t wbytes = (size_t)((g.oc + 31) / 32) * (g.c / 16) * 9 * 3 * 1024 + (oct == 64 ? (size_t)(g.c / 16) * 9 * 3 * 1024 : 0);
        void* dwf = nullptr;
        const bool cacheable = wt->mem == LELE_MEM_WEIGHT;
        auto key = std::make_tuple((const void*)wt->data, wbytes, 330 + 9);
        auto it = cacheable ? ctx->weights.find(key) : ctx->weights.end();
        if (it != ctx->weights.end()) {
            dwf = it->second;
        } else {
            if (cacheable) {
                LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
                LELE_HIP_CHECK(hipMalloc(&dwf, wbytes));
                ctx->weights[key] = dwf;
            } else {
                LELE_TRY(ctx->arena_alloc(wbytes, &dwf));
            }
            const int oc_pad = ((g.oc + oct - 1) / oct) * oct;
            LELE_HIP_CHECK(hipMemsetAsync(dwf, 0, wbytes, ctx->stream));
            hipLaunchKernelGGL(conv_wfrag_kernel, dim3(grid_for((int64_t)(oc_pad / 32) * (g.c / 16) * 9 * 64)), dim3(256), 0, ctx->stream, dw,
                               (cu32x4*)dwf, g.oc, g.c, 9);
        }
        ConvEpi epi{out, db, g, act};
        const WinTile tile = pick_win_tile(g.ow, g.oh, 128, 3, 2, C3S2::POS, (int64_t)g.n * ((g.oc + oct - 1) / oct), ctx->num_cus);
        const int ntiles = tile.tiles_x * ((g.oh + tile.th - 1) / tile.th), nblocks = (g.oc + oct - 1) / oct;
        const int osplit = (int64_t)g.n * ntiles >= 2 * (int64_t)ctx->num_cus ? 1 : nblocks, nocb = nblocks / osplit;
        const int64_t items = (int64_t)g.n * ntiles * osplit;
        LELE_REQUIRE(items < (int64_t(1) << 31), "conv2d: more than 2^31 tiles");
        const dim3 pgrid((unsigned)std::min<int64_t>(items, conv_env("LELE_HIP_CONV_WG_PER_CU", 2) * (int64_t)ctx->num_cus));
        if (oct == 64) {
            auto kern = conv_window_s2_kernel<64>;
            LELE_HIP_CHECK(lele::ensure_dyn_lds(reinterpret_cast<const void*>(kern), C3S2::STAGE));
            hipLaunchKernelGGL(kern, pgrid, dim3(512), C3S2::STAGE, ctx->stream, dx, (const cu32x4*)dwf, epi, tile, ntiles, nocb, osplit, (int)items);
        } else {
            auto kern = conv_window_s2_kernel<32>;
            LELE_HIP_CHECK(lele::ensure_dyn_lds(reinterpret_cast<const void*>(kern), C3S2::STAGE));
            hipLaunchKernelGGL(kern, pgrid, dim3(512), C3S2::STAGE, ctx->stream, dx, (const cu32x4*)dwf, epi, tile, ntiles, nocb, osplit, (int)items);
        }
    } else if (g.group == 1 && g.kh == 3 && g.kw == 3 && g.dh == 1 && g.dw == 1 && g.sh == g.sw && (g.sh == 1 || g.sh == 2) && g.oc <= 16 &&
               g.c <= 64 && g.ow >= 16 && g.n <= 65535 &&
               (int64_t)g.n * ((g.ow + 31) / 32) * ((g.oh + 7) / 8) >= 2 * (int64_t)ctx->num_cus) {
        // few channels over a batch: the direct kernel (see conv3x3_direct_kernel).  Measured on the Yolo-shaped network at batch 64
        // against the implicit GEMM: 3 -> 16 stride 2 at 320 x 320: 227 against 579 us, 16 -> 8 at 160 x 160: 88 against 251,
        // 8 -> 16: ~95 against 180; with 32 output channels the MFMA tile is full and the GEMM wins (16 -> 32 stride 2: 322 against
        // 596), and one image does not fill the chip with 8 x 32 tiles -- both stay on the GEMM
        const int ocb = g.oc <= 8 ? 8 : 16;
        const size_t wbytes = (size_t)g.c * 9 * ocb * 4;
        void* dwq = nullptr;
        const bool cacheable = wt->mem == LELE_MEM_WEIGHT;
        auto key = std::make_tuple((const void*)wt->data, wbytes, 310 + ocb);
        auto it = cacheable ? ctx->weights.find(key) : ctx->weights.end();
        if (it != ctx->weights.end()) {
            dwq = it->second;
        } else {
            if (cacheable) {
                LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
                LELE_HIP_CHECK(hipMalloc(&dwq, wbytes));
                ctx->weights[key] = dwq;
            } else {
                LELE_TRY(ctx->arena_alloc(wbytes, &dwq));
            }
            hipLaunchKernelGGL(conv3x3_wperm_kernel, dim3(grid_for((int64_t)g.c * 9 * ocb)), dim3(256), 0, ctx->stream, dw, (float*)dwq, g.oc, g.c, ocb);
        }
        const int s_ = g.sh, ph = 7 * s_ + 3, pp = 31 * s_ + 3 + 1;
        int icc = g.c;
        while ((size_t)icc * ph * pp * 4 > 36 * 1024) icc = (icc + 1) / 2;  // the window of `icc` channels per LDS pass: four workgroups a CU
        const size_t lds = (size_t)icc * ph * pp * 4;
        const int tiles_x = (g.ow + 31) / 32, tiles_y = (g.oh + 7) / 8;
        const dim3 dgrid((unsigned)(tiles_x * tiles_y), (unsigned)g.n);
#define LELE_C3(OCB_, S_)                                                                                            \
    do {                                                                                                             \
        auto kern = conv3x3_direct_kernel<OCB_, S_>;                                                                  \
        if (lds > 60 * 1024) LELE_HIP_CHECK(lele::ensure_dyn_lds(reinterpret_cast<const void*>(kern), (int)lds));     \
        hipLaunchKernelGGL(kern, dgrid, dim3(256), lds, ctx->stream, dx, (const float*)dwq, db, out, g, act, tiles_x, icc); \
    } while (0)
        if (s_ == 1) {
            if (ocb == 8) LELE_C3(8, 1);
            else LELE_C3(16, 1);
        } else {
            if (ocb == 8) LELE_C3(8, 2);
            else LELE_C3(16, 2);
        }
#undef LELE_C3
    } else {
        ConvWLoad al{dw, g, (int)((((uintptr_t)dw & 15) == 0) && g.K % 4 == 0)};
        ConvEpi epi{out, db, g, act};
        const int64_t img_elems = (int64_t)g.icg * g.ih * g.iw;
        LELE_REQUIRE(img_elems < (int64_t(1) << 31), "conv2d: one image group exceeds 2^31 elements");
        if (g.kh == 1 && g.kw == 1 && g.sh == 1 && g.sw == 1 && g.pt == 0 && g.pl == 0 && g.oh == g.ih && g.ow == g.iw) {
            // pointwise: B[p][k] = x[img][grp*ICg + k][p] is a plain column-major operand; batch b = img*G + grp
            gemm::LoadKRow bl{dx, g.group == 1 ? (int64_t)g.xbs : img_elems, (int64_t)g.plane, g.plane, g.K};
            gemm::launch(ctx->stream, al, bl, epi, g.ocg, g.plane, g.K, g.n * g.group, ctx->num_cus);
        } else if (g.icg % 4 == 0) {
            // tap-major K: weights permuted to [OC][tap][ic] once (cached when the caller declared them immutable)
            const int khw = g.kh * g.kw;
            const size_t wbytes = (size_t)g.oc * g.K * 4;
            void* dwt = nullptr;
            const bool cacheable = wt->mem == LELE_MEM_WEIGHT;
            auto key = std::make_tuple((const void*)wt->data, wbytes, 301);
            auto it = cacheable ? ctx->weights.find(key) : ctx->weights.end();
            if (it != ctx->weights.end()) {
                dwt = it->second;
            } else {
                if (cacheable) {
                    LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
                    LELE_HIP_CHECK(hipMalloc(&dwt, wbytes));
                    ctx->weights[key] = dwt;
                } else {
                    LELE_TRY(ctx->arena_alloc(wbytes, &dwt));
                }
                hipLaunchKernelGGL(conv_wperm_kernel, dim3(grid_for((int64_t)g.oc * g.K)), dim3(256), 0, ctx->stream, dw,
                                   (float*)dwt, g.oc, g.icg, khw);
            }
            ConvWLoad alt{(const float*)dwt, g, 1};  // hipMalloc / arena chunks are 16-B aligned and K % 4 == 0
            ConvXLoadTap bl{dx, g, make_fastdiv(g.ow, g.plane), make_fastdiv(g.icg, g.K), make_fastdiv(g.kw, khw)};
            gemm::launch(ctx->stream, alt, bl, epi, g.ocg, g.plane, g.K, g.n * g.group, ctx->num_cus);
        } else {
            ConvXLoad bl{dx, g, make_fastdiv(g.ow, g.plane), make_fastdiv(g.kh * g.kw, g.K), make_fastdiv(g.kw, g.kh * g.kw)};
            gemm::launch(ctx->stream, al, bl, epi, g.ocg, g.plane, g.K, g.n * g.group, ctx->num_cus);
        }
    }
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- ConvInteger family (conv2d.rs:1507-2761).  On x86 lele subtracts the zero points in f32 and runs its f32 GEMM
// (conv2d_with_zero_points, :1507-2000: w - w_zp at :1631-1718, x - x_zp inside im2col_with_zp), so the result is the f32
// convolution of the centred operands -- where a PADDED cell is the u8 value 0, i.e. -x_zp once centred (:2025 "pad value is
// (0 - x_zp)", :2043-2050, :2183-2199), not the zero point.
// Here (group == 1, integer zero points, u8-coded operands): the codes themselves on the i8 matrix cores.  With x~ the u8 image
// padded with zeros, a = x~ - 128 and b = w - 128 as i8:
//     sum (x~ - zx)(w - zw) = sum a b + (128 - zw) sum x~ + (128 - zx) sum w - 16384 K + K zx zw          (K = C kh kw)
// every term an exact i32; `sum x~` over a window is a 9-tap sum of the per-pixel channel sums T, `sum w` one number per output
// channel.  The image is re-laid once as u8 [N, H, W, Cp] (Cp = C rounded up to 32; the same pass quantises an f32 source with
// lele's DynamicQuantizeLinear formula and produces T): 16 consecutive channels of a pixel are then ONE 16-byte load -- a B
// fragment of v_mfma_i32_32x32x32_i8 straight from L2 through a buffer resource, a position outside the image an offset past the
// resource (reads 0 = the padded u8 value) -- and u8 -> i8 is a flip of the top bit of every byte.  No LDS, no barrier; the four
// waves of a workgroup share one block of 32 output channels (its weight fragments stay in L1) and 128 positions each.
// Everything else (groups, fractional zero points, device-resident weights) centres in f32 as lele does -- on an image padded
// with zeros FIRST.
typedef int ci_v4i __attribute__((ext_vector_type(4)));
typedef int ci_v16i __attribute__((ext_vector_type(16)));
struct Ci8Prm {          // zero points: host values (conv_integer) or the device record of the dynamic quantisation (from_f32)
    const lele::QParamsDev* dq;
    int zx_host, zw;
};
// u8 image [N, H, W, Cp] + per-pixel channel sums T [N, H, W] from f32 sources [N, src_c, H*W] landing in channels [ch_off, ..):
// QUANT: q = clamp(round(x * inv_scale + zp), 0, 255) (conv2d.rs:2388-2394, f32::round = half away from zero); else the value is
// already a u8 code.  One thread per (image, pixel, 4 channels); T is accumulated with integer atomics (order-free).
template <bool QUANT>
__global__ __launch_bounds__(256) void ci8_pack_kernel(const float* __restrict__ src, int64_t n_img, int src_c, int ch_off, int cp, int64_t spatial,
                                                       const lele::QParamsDev* __restrict__ prm, unsigned* __restrict__ img, int* __restrict__ tsum) {
    const int quads = (src_c + 3) / 4;
    const int64_t total = n_img * quads * spatial;
    float inv_scale = 1.0f, zp = 0.0f;
    if (QUANT) inv_scale = prm[0].inv_scale, zp = prm[0].zp;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t pix = i % spatial, r = i / spatial;
        const int q = (int)(r % quads);
        const int64_t n = r / quads;
        unsigned word = 0u;
        int sum = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * q + e;
            float v = c < src_c ? src[(n * src_c + c) * spatial + pix] : 0.0f;
            if (QUANT && c < src_c) {
                v = roundf(v * inv_scale + zp);
                v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
            }
            const unsigned code = (unsigned)(int)v & 255u;
            word |= code << (8 * e);
            sum += (int)code;
        }
        // ch_off is a multiple of 4 for every source but possibly the last of a multi-source concat: bytes are placed one by one then
        unsigned char* dst = reinterpret_cast<unsigned char*>(img) + (n * spatial + pix) * cp + ch_off + 4 * q;
        if ((ch_off & 3) == 0) *reinterpret_cast<unsigned*>(dst) = word;
        else
            for (int e = 0; e < 4 && 4 * q + e < src_c; ++e) dst[e] = (unsigned char)(word >> (8 * e));
        atomicAdd(&tsum[n * spatial + pix], sum);
    }
}
// weights [OC][C][taps] (u8 codes as f32) -> i8 fragments b = w - 128 in MFMA order [oc tile][tap][chunk][64 lanes] x 16 bytes (lane
// (l31, hv): output channel 32 tile + l31, input channels 32 chunk + 16 hv + [0, 16); zero beyond OC / C) and wsum[oc] = sum w
__global__ __launch_bounds__(256) void ci8_wfrag_kernel(const float* __restrict__ w, int oc, int c, int taps, int chunks, ci_v4i* __restrict__ frag,
                                                        int* __restrict__ wsum) {
    const int64_t total = (int64_t)((oc + 31) / 32) * taps * chunks * 64;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int lane = (int)(i & 63), chunk = (int)((i >> 6) % chunks);
        const int64_t r = (i >> 6) / chunks;
        const int tap = (int)(r % taps), tile = (int)(r / taps);
        const int o = tile * 32 + (lane & 31), c0 = chunk * 32 + 16 * (lane >> 5);
        ci_v4i v;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            unsigned word = 0u;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ch = c0 + 4 * d + e;
                const int code = o < oc && ch < c ? (int)w[((int64_t)o * c + ch) * taps + tap] : 128;
                word |= ((unsigned)(code - 128) & 255u) << (8 * e);
            }
            v[d] = (int)word;
        }
        frag[i] = v;
    }
    for (int o = blockIdx.x * 256 + threadIdx.x; o < oc; o += gridDim.x * 256) {
        int sum = 0;
        for (int64_t k = 0; k < (int64_t)c * taps; ++k) sum += (int)w[(int64_t)o * c * taps + k];
        wsum[o] = sum;
    }
}
// the convolution: 256 threads = four waves; a wave multiplies 32 output channels x NJ strips of 32 positions.  KSPLIT = false: the
// waves of a workgroup own different positions (4 NJ x 32 a workgroup); KSPLIT = true (small layers: not enough tiles to go round):
// they share ONE tile, take every fourth (tap, chunk) step each and meet in LDS -- integer sums, any order.  The operands of the next
// step are requested before the products of this one (two register sets): a step is one L2 round trip otherwise.
// grid (position blocks, ceil(OC / 32), N).
template <int NJ, bool KSPLIT>
__global__ __launch_bounds__(256) void ci8_conv_kernel(const unsigned char* __restrict__ img, const int* __restrict__ tsum, const ci_v4i* __restrict__ frag,
                                                       const int* __restrict__ wsum, Ci8Prm prm, ConvGeom g, int cp, float* __restrict__ out) {
    __shared__ int red[KSPLIT ? 3 * NJ * 16 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hv = lane >> 5, l31 = lane & 31;
    const int tile = blockIdx.y, n = blockIdx.z, chunks = cp / 32, taps = g.kh * g.kw, steps = taps * chunks;
    const int p0 = (KSPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave) * (32 * NJ);
    if (!KSPLIT && p0 >= g.plane) return;
    const unsigned char* base = img + (int64_t)n * g.ih * g.iw * cp;
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, (short)0, (int)((unsigned)g.ih * g.iw * cp), 0x00020000);
    int iy0[NJ], ix0[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int p = p0 + 32 * j + l31;
        p = p < g.plane ? p : g.plane - 1;
        const int oy = p / g.ow;
        iy0[j] = oy * g.sh - g.pt;
        ix0[j] = (p - oy * g.ow) * g.sw - g.pl;
    }
    ci_v16i acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0;
    const ci_v4i* fr = frag + (int64_t)tile * steps * 64 + lane;
    auto load = [&](int t, ci_v4i& af, ci_v4i (&bf)[NJ]) {  // step t = (tap, chunk of 32 channels)
        const int tap = t / chunks, ch = t - tap * chunks, a = tap / g.kw, b = tap - a * g.kw;
        af = fr[(int64_t)t * 64];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int iy = iy0[j] + a * g.dh, ix = ix0[j] + b * g.dw;
            const bool in = iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw;
            const unsigned off = in ? (unsigned)(iy * g.iw + ix) * (unsigned)cp + 16u * hv + 32u * ch : 0x80000000u;  // past the resource (< 2^31 bytes): reads 0
            bf[j] = __builtin_bit_cast(ci_v4i, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0));
        }
    };
    auto mult = [&](const ci_v4i& af, ci_v4i (&bf)[NJ]) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            bf[j] ^= (int)0x80808080u;  // u8 -> i8: x - 128
            acc[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(af, bf[j], acc[j], 0, 0, 0);
        }
    };
    const int stride = KSPLIT ? 4 : 1;
    ci_v4i a0, a1, b0[NJ], b1[NJ];
    int t = KSPLIT ? wave : 0;
    if (t < steps) load(t, a0, b0);
    while (t < steps) {
        if (t + stride < steps) load(t + stride, a1, b1);
        mult(a0, b0);
        t += stride;
        if (t >= steps) break;
        if (t + stride < steps) load(t + stride, a0, b0);
        mult(a1, b1);
        t += stride;
    }
    if (KSPLIT) {
        if (wave > 0) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((wave - 1) * NJ + j) * 16 + r) * 64 + lane] = acc[j][r];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] += red[((w * NJ + j) * 16 + r) * 64 + lane];
    }
    const int zx = prm.dq ? prm.dq[0].zp_i : prm.zx_host, zw = prm.zw, K = g.c * taps;
    const int konst = K * zx * zw - 16384 * K;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        int sx = 0;  // sum of the u8 codes over this position's window (zeros outside the image): taps of the per-pixel channel sums
        for (int tap = 0; tap < taps; ++tap) {
            const int a = tap / g.kw, b = tap - a * g.kw, iy = iy0[j] + a * g.dh, ix = ix0[j] + b * g.dw;
            if (iy >= 0 && iy < g.ih && ix >= 0 && ix < g.iw) sx += tsum[(int64_t)n * g.ih * g.iw + iy * g.iw + ix];
        }
        const int p = p0 + 32 * j + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hv;
            if (p < g.plane && o < g.oc)
                out[((int64_t)n * g.oc + o) * g.plane + p] = (float)(acc[j][r] + (128 - zw) * sx + (128 - zx) * wsum[o] + konst);
        }
    }
}
// the f32 route's image: x (or its quantised codes) centred, inside a border of -zp cells (the padded u8 value 0)
__global__ void ci_pad_center_kernel(const float* __restrict__ x, int64_t planes, int ih, int iw, int pt, int pl, int ph, int pw, float zp_host,
                                     const lele::QParamsDev* __restrict__ dq, float* __restrict__ out) {
    const float zp = dq ? dq[0].zp : zp_host;
    const int64_t total = planes * ph * pw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t pl_ = i / ((int64_t)ph * pw);
        const int r = (int)(i - pl_ * ph * pw), y = r / pw - pt, xx = r % pw - pl;
        const bool in = y >= 0 && y < ih && xx >= 0 && xx < iw;
        out[i] = (in ? x[(pl_ * ih + y) * iw + xx] : 0.0f) - zp;
    }
}
__global__ void ci_sub_kernel(const float* __restrict__ x, int64_t n, float zp, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = x[i] - zp;
}
// conv_integer_from_f32 (:2246-2418, x86 branch :2388-2394): q = clamp(round(x * inv_scale + zp), 0, 255); the conv sees q - zp.
// src is [N, src_c, spatial]; it lands in channels [ch_off, ch_off + src_c) of a [N, total_c, spatial] tensor (multi form)
// (the codes themselves, q, not q - zp: ci_pad_center_kernel centres them together with the border)
__global__ void ci_quant_codes_kernel(const float* __restrict__ src, int64_t n_img, int64_t src_c, int64_t total_c,
                                      int64_t ch_off, int64_t spatial, const QParamsDev* __restrict__ prm,
                                      float* __restrict__ out) {
    const QParamsDev q = prm[0];
    const int64_t per_img = src_c * spatial, total = n_img * per_img;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t img = i / per_img, r = i - img * per_img;
        float v = roundf(src[i] * q.inv_scale + q.zp);  // f32::round: half away from zero; mul then add, not fused
        v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
        out[(img * total_c + ch_off) * spatial + r] = v;
    }
}
__global__ void ci_scale_out_kernel(const QParamsDev* __restrict__ prm, float* __restrict__ scale) { scale[0] = prm[0].scale; }
// fused_scale_bias / fused_scale_bias_silu (:2636-2761, x86 branches): x = d*scale + bias[c]; silu: x / (1 + exp(-x))
__global__ void ci_scale_bias_kernel(const float* __restrict__ d, int64_t total, int64_t channels, int64_t spatial,
                                     const float* __restrict__ scale_dev, float scale_mul, const float* __restrict__ bias,
                                     int silu, float* __restrict__ out) {
    const float scale = scale_dev ? scale_dev[0] * scale_mul : scale_mul;  // in_scale * w_scale, as the caller multiplies
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = (i / spatial) % channels;
        const float x = d[i] * scale + bias[c];
        out[i] = silu ? x / (1.0f + expf(-x)) : x;
    }
}

int conv_geom_from(const LeleTensor* x, const int64_t* xshape, const LeleTensor* w, const int64_t* dilations, size_t ndil,
                   int64_t group, const int64_t* pads, size_t npads, const int64_t* strides, size_t nstr, ConvGeom* out) {
    ConvGeom g{};
    g.n = (int)xshape[0];
    g.c = (int)xshape[1];
    g.ih = (int)xshape[2];
    g.iw = (int)xshape[3];
    g.oc = (int)w->shape[0];
    g.kh = (int)w->shape[2];
    g.kw = (int)w->shape[3];
    g.group = (int)group;
    LELE_REQUIRE(group >= 1 && g.c % g.group == 0 && g.oc % g.group == 0, "conv_integer: channels not divisible by group");
    g.icg = g.c / g.group;
    g.ocg = g.oc / g.group;
    LELE_REQUIRE(w->shape[1] == g.icg, "conv_integer: weight C_in/g mismatch");
    g.dh = (int)attr(dilations, ndil, 0, 1);
    g.dw = (int)attr(dilations, ndil, 1, 1);
    g.sh = (int)attr(strides, nstr, 0, 1);
    g.sw = (int)attr(strides, nstr, 1, 1);
    int pb, pr;
    if (npads >= 4) {
        g.pt = (int)pads[0];
        g.pl = (int)pads[1];
        pb = (int)pads[2];
        pr = (int)pads[3];
    } else if (npads >= 2) {
        g.pt = pb = (int)pads[0];
        g.pl = pr = (int)pads[1];
    } else {
        g.pt = g.pl = pb = pr = 0;
    }
    const int64_t nh = (int64_t)g.ih + g.pt + pb - (int64_t)g.dh * (g.kh - 1) - 1;
    const int64_t nw = (int64_t)g.iw + g.pl + pr - (int64_t)g.dw * (g.kw - 1) - 1;
    LELE_REQUIRE(nh >= 0 && nw >= 0 && g.sh > 0 && g.sw > 0,
                 "conv2d_with_zero_points: output dimensions must be positive");  // conv2d.rs:1602
    g.oh = (int)(nh / g.sh + 1);
    g.ow = (int)(nw / g.sw + 1);
    g.K = g.icg * g.kh * g.kw;
    g.plane = g.oh * g.ow;
    *out = g;
    return 0;
}

float host_scalar(const LeleTensor* t) {  // zp.data[0] or 0 (conv2d.rs:2227-2236); zero points are host-side attributes
    if (!t || numel(t) == 0 || t->mem == LELE_MEM_DEVICE) return 0.0f;
    return *(const float*)t->data;
}

// centred weights (w - w_zp), cached for immutable weights
int centred_weights(LeleCtx* ctx, const LeleTensor* w, float w_zp, const float** out) {
    const void* dw = nullptr;
    if (w_zp == 0.0f) {
        LELE_TRY(ctx->dev_ptr(w, &dw));
        *out = (const float*)dw;
        return 0;
    }
    const size_t bytes = (size_t)numel(w) * 4;
    int zbits;
    memcpy(&zbits, &w_zp, 4);
    auto key = std::make_tuple((const void*)w->data, bytes, 500 + (zbits & 0x7fffff));
    const bool cacheable = w->mem == LELE_MEM_WEIGHT;
    if (cacheable) {
        auto it = ctx->weights.find(key);
        if (it != ctx->weights.end()) {
            *out = (const float*)it->second;
            return 0;
        }
    }
    LELE_TRY(ctx->dev_ptr(w, &dw));
    void* adj = nullptr;
    if (cacheable) {
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMalloc(&adj, std::max<size_t>(bytes, 16)));
        ctx->weights[key] = adj;
    } else {
        LELE_TRY(ctx->arena_alloc(std::max<size_t>(bytes, 16), &adj));
    }
    hipLaunchKernelGGL(ci_sub_kernel, dim3(grid_for(numel(w))), dim3(256), 0, ctx->stream, (const float*)dw, numel(w), w_zp,
                       (float*)adj);
    *out = (const float*)adj;
    return 0;
}

inline bool steps_lt8(int steps) { return steps < 8; }  // too few (tap, chunk) steps to share among four waves
// One ConvInteger: `nsrc` f32 sources [N, c_i, H, W] concatenated along C (u8 codes, or -- dq != NULL -- values to be quantised with
// the parameters at dq), weights w (u8 codes as f32), zero points zx_host (ignored when dq) and w_zp.  See the section comment.
int ci_run(LeleCtx* ctx, ConvGeom g, const LeleTensor* w, float w_zp, const float* const* srcs, const int64_t* src_c, int nsrc, const QParamsDev* dq,
           float zx_host, float* out) {
    if ((int64_t)g.n * g.oc * g.plane == 0) return 0;
    const int taps = g.kh * g.kw;
    const int64_t spatial = (int64_t)g.ih * g.iw, K = (int64_t)g.c * taps;
    auto code = [](float v) { return v >= 0.0f && v <= 255.0f && v == (float)(int)v; };
    bool i8 = g.group == 1 && code(w_zp) && (dq || code(zx_host)) && K * 65025 < (int64_t(1) << 31) && w->mem != LELE_MEM_DEVICE &&
              spatial * (((int64_t)g.c + 31) & ~int64_t(31)) < (int64_t(1) << 31) - 512 && g.n <= 65535 && (g.oc + 31) / 32 <= 65535 &&
              !lab_env("LELE_HIP_CONV_INTEGER_F32");
    const int cp = (g.c + 31) & ~31, chunks = cp / 32, tiles = (g.oc + 31) / 32;
    const size_t fbytes = (size_t)tiles * taps * chunks * 1024, wbytes = fbytes + (size_t)g.oc * 4;
    const bool cacheable = w->mem == LELE_MEM_WEIGHT;
    const auto key = std::make_tuple((const void*)w->data, wbytes, 520);
    const auto key_not = std::make_tuple((const void*)w->data, wbytes, 521);  // verdict "not u8 codes" of an immutable tensor (a 16-byte marker)
    void* fw = nullptr;
    if (i8) {  // the weights must be u8 codes: a host array, scanned once per immutable tensor (every call otherwise)
        auto it = cacheable ? ctx->weights.find(key) : ctx->weights.end();
        if (it != ctx->weights.end()) {
            fw = it->second;
        } else if (cacheable && ctx->weights.count(key_not)) {
            i8 = false;
        } else {
            const float* hw = (const float*)w->data;
            const int64_t nw = numel(w);
            for (int64_t i = 0; i < nw && i8; ++i) i8 = code(hw[i]);
            if (!i8 && cacheable && !ctx->capturing) {
                void* marker = nullptr;
                LELE_HIP_CHECK(hipMalloc(&marker, 16));
                ctx->weights[key_not] = marker;
            }
        }
    }
    if (i8) {
        if (!fw) {
            const void* dw = nullptr;
            LELE_TRY(ctx->dev_ptr(w, &dw));
            if (cacheable) {
                LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
                LELE_HIP_CHECK(hipMalloc(&fw, wbytes));
                ctx->weights[key] = fw;
            } else {
                LELE_TRY(ctx->arena_alloc(wbytes, &fw));
            }
            hipLaunchKernelGGL(ci8_wfrag_kernel, dim3(grid_for((int64_t)tiles * taps * chunks * 64)), dim3(256), 0, ctx->stream, (const float*)dw, g.oc, g.c,
                               taps, chunks, (ci_v4i*)fw, (int*)((char*)fw + fbytes));
        }
        void *img = nullptr, *ts = nullptr;
        LELE_TRY(ctx->arena_alloc((size_t)g.n * spatial * cp + 512, &img));
        LELE_TRY(ctx->arena_alloc((size_t)g.n * spatial * 4, &ts));
        LELE_HIP_CHECK(hipMemsetAsync(ts, 0, (size_t)g.n * spatial * 4, ctx->stream));
        int ch_off = 0;
        for (int i = 0; i < nsrc; ++i) {
            const int64_t work = (int64_t)g.n * ((src_c[i] + 3) / 4) * spatial;
            if (work && dq)
                hipLaunchKernelGGL(ci8_pack_kernel<true>, dim3(grid_for(work)), dim3(256), 0, ctx->stream, srcs[i], (int64_t)g.n, (int)src_c[i], ch_off, cp,
                                   spatial, dq, (unsigned*)img, (int*)ts);
            else if (work)
                hipLaunchKernelGGL(ci8_pack_kernel<false>, dim3(grid_for(work)), dim3(256), 0, ctx->stream, srcs[i], (int64_t)g.n, (int)src_c[i], ch_off, cp,
                                   spatial, dq, (unsigned*)img, (int*)ts);
            ch_off += (int)src_c[i];
        }
        const Ci8Prm prm{dq, (int)zx_host, (int)w_zp};
        // the widest wave tile that still gives every CU two workgroups; below that the waves of a workgroup split K instead
        const int64_t units = (int64_t)g.n * tiles, want = 2 * (int64_t)ctx->num_cus;
        auto blocks = [&](int nj, bool ks) { return ((int64_t)g.plane + (ks ? 1 : 4) * 32 * nj - 1) / ((ks ? 1 : 4) * 32 * nj); };
#define LELE_CI8(NJ_, KS_)                                                                                                          \
    hipLaunchKernelGGL((ci8_conv_kernel<NJ_, KS_>), dim3((unsigned)blocks(NJ_, KS_), (unsigned)tiles, (unsigned)g.n), dim3(256), 0, ctx->stream, \
                       (const unsigned char*)img, (const int*)ts, (const ci_v4i*)fw, (const int*)((char*)fw + fbytes), prm, g, cp, out)
        if (units * blocks(4, false) >= want) LELE_CI8(4, false);
        else if (units * blocks(2, false) >= want) LELE_CI8(2, false);
        else if (units * blocks(1, false) >= want || steps_lt8(taps * chunks)) LELE_CI8(1, false);
        else if (units * blocks(2, true) >= want) LELE_CI8(2, true);
        else LELE_CI8(1, true);
#undef LELE_CI8
        LELE_HIP_CHECK(hipGetLastError());
        return 0;
    }
    // ---- f32: codes -> zero-padded, centred image -> lele's f32 convolution without padding
    const float* dwc = nullptr;
    LELE_TRY(centred_weights(ctx, w, w_zp, &dwc));
    const float* codes = srcs[0];
    if (dq || nsrc > 1) {
        void* q = nullptr;
        LELE_TRY(ctx->arena_alloc((size_t)std::max<int64_t>((int64_t)g.n * g.c * spatial, 1) * 4, &q));
        int64_t ch_off = 0;
        for (int i = 0; i < nsrc; ++i) {
            const int64_t len = (int64_t)g.n * src_c[i] * spatial;
            LELE_REQUIRE(dq, "conv_integer: several sources need the dynamic quantisation");
            if (len)
                hipLaunchKernelGGL(ci_quant_codes_kernel, dim3(grid_for(len)), dim3(256), 0, ctx->stream, srcs[i], (int64_t)g.n, src_c[i], (int64_t)g.c,
                                   ch_off, spatial, dq, (float*)q);
            ch_off += src_c[i];
        }
        codes = (const float*)q;
    }
    const int pb = (g.oh - 1) * g.sh + g.dh * (g.kh - 1) + 1 - g.ih - g.pt, pr = (g.ow - 1) * g.sw + g.dw * (g.kw - 1) + 1 - g.iw - g.pl;
    const int ph = g.ih + g.pt + std::max(pb, 0), pw = g.iw + g.pl + std::max(pr, 0);  // (rows / columns no window reaches are not needed)
    void* xc = nullptr;
    const int64_t planes = (int64_t)g.n * g.c;
    LELE_TRY(ctx->arena_alloc((size_t)std::max<int64_t>(planes * ph * pw, 1) * 4, &xc));
    hipLaunchKernelGGL(ci_pad_center_kernel, dim3(grid_for(planes * ph * pw)), dim3(256), 0, ctx->stream, codes, planes, g.ih, g.iw, g.pt, g.pl, ph, pw,
                       zx_host, dq, (float*)xc);
    ConvGeom gp = g;
    gp.ih = ph, gp.iw = pw, gp.pt = gp.pl = 0;
    LeleTensor wv = *w;
    wv.mem = LELE_MEM_DEVICE;  // the centred copy is what run_conv2d sees (its own weight cache keys on the pointer below)
    wv.data = dwc;
    return run_conv2d(ctx, &wv, (const float*)xc, dwc, nullptr, gp, LELE_ACT_NONE, out);
}

}  // namespace

extern "C" {

static int conv2d_entry(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                        const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                        const int64_t* strides, size_t nstr, int act, const LelePitch* pv, LeleBuf* out, int64_t* out_shape,
                        int32_t* out_rank, const LeleTensor* res = nullptr) {
    LELE_REQUIRE(ctx && x && w && out, "conv2d: NULL argument");
    LELE_REQUIRE(x->rank == 4, "Conv2d: expected rank-4 input [N,C,H,W], got rank %d", x->rank);        // conv2d.rs:196
    LELE_REQUIRE(w->rank == 4, "Conv2d: expected rank-4 weight [C_out,C_in/g,kH,kW], got rank %d", w->rank);
    LELE_REQUIRE(x->dtype == LELE_F32 && w->dtype == LELE_F32, "conv2d: f32 tensors required");
    LELE_REQUIRE(act >= LELE_ACT_NONE && act <= LELE_ACT_SILU, "conv2d: unknown activation %d", act);
    if (act == LELE_ACT_SILU) {  // read per call, like the other run-time switches
        const char* e = getenv("LELE_HIP_CONV_SILU_EXACT");
        if (e && *e && strcmp(e, "0") != 0) act = kActSiluExact;
    }
    LELE_REQUIRE(group >= 1, "conv2d: group must be >= 1");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    ConvGeom g{};
    g.n = (int)x->shape[0];
    g.c = (int)x->shape[1];
    g.ih = (int)x->shape[2];
    g.iw = (int)x->shape[3];
    g.oc = (int)w->shape[0];
    g.kh = (int)w->shape[2];
    g.kw = (int)w->shape[3];
    g.group = (int)group;
    LELE_REQUIRE(g.c % g.group == 0 && g.oc % g.group == 0, "conv2d: channels not divisible by group");
    g.icg = g.c / g.group;
    g.ocg = g.oc / g.group;
    LELE_REQUIRE(w->shape[1] == g.icg, "conv2d: weight C_in/g = %lld but input has %d channels per group",
                 (long long)w->shape[1], g.icg);
    g.dh = (int)attr(dilations, ndil, 0, 1);
    g.dw = (int)attr(dilations, ndil, 1, 1);
    g.sh = (int)attr(strides, nstr, 0, 1);
    g.sw = (int)attr(strides, nstr, 1, 1);
    int pb, pr;
    if (npads >= 4) {  // [top, left, bottom, right], conv2d.rs:246-273
        g.pt = (int)pads[0];
        g.pl = (int)pads[1];
        pb = (int)pads[2];
        pr = (int)pads[3];
    } else if (npads >= 2) {
        g.pt = pb = (int)pads[0];
        g.pl = pr = (int)pads[1];
    } else {
        g.pt = g.pl = pb = pr = 0;
    }
    const int64_t nh = (int64_t)g.ih + g.pt + pb - (int64_t)g.dh * (g.kh - 1) - 1;
    const int64_t nw = (int64_t)g.iw + g.pl + pr - (int64_t)g.dw * (g.kw - 1) - 1;
    LELE_REQUIRE(nh >= 0 && nw >= 0 && g.sh > 0 && g.sw > 0, "conv2d: output dimensions must be positive");  // :286
    g.oh = (int)(nh / g.sh + 1);
    g.ow = (int)(nw / g.sw + 1);
    g.K = g.icg * g.kh * g.kw;
    ctx->conv_calls += 1;  // reset_conv_stats / print_conv_stats (conv2d.rs:75,101)
    ctx->conv_macs += (int64_t)g.n * g.oc * g.oh * g.ow * g.K;
    g.plane = g.oh * g.ow;
    if (bias) LELE_REQUIRE(numel(bias) >= g.oc, "conv2d: bias has %lld entries for %d channels", (long long)numel(bias), g.oc);
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dwp = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(w, &dwp));
    if (bias) LELE_TRY(ctx->dev_ptr(bias, &db));
    float* dst = nullptr;
    if (res) {  // out = act(conv + bias) + res
        LELE_REQUIRE(res->dtype == LELE_F32 && res->rank == 4 && res->shape[0] == g.n && res->shape[1] == g.oc && res->shape[2] == g.oh &&
                         res->shape[3] == g.ow, "conv2d_res: the residual must be an f32 tensor of the result's shape [%d,%d,%d,%d]", g.n, g.oc, g.oh, g.ow);
        LELE_REQUIRE((int64_t)g.oc * g.plane < (int64_t(1) << 32) && g.n <= 65535, "conv2d_res: an image exceeds 2^32 elements or more than 65535 images");
        const void* dr = nullptr;
        LELE_TRY(ctx->dev_ptr(res, &dr));
        g.res = (const float*)dr;
        g.rbs = (int64_t)g.oc * g.plane;
        if (pv && pv->y_pitch) {
            LELE_REQUIRE(res->mem == LELE_MEM_DEVICE && pv->y_pitch >= g.rbs, "conv2d_res: y_pitch needs a device tensor and must cover one image");
            g.rbs = pv->y_pitch;
        }
    }
    if (pv) {  // channel views: see LelePitch in lele_hip.h
        LELE_REQUIRE(pv->y_pitch == 0 || res, "conv2d_pitched: y_pitch must be 0 (one tensor operand)");
        LELE_REQUIRE(pv->x_pitch == 0 || (x->mem == LELE_MEM_DEVICE && pv->x_pitch >= (int64_t)g.c * g.ih * g.iw),
                     "conv2d_pitched: x_pitch needs a device tensor and must cover one image");
        g.xbs = pv->x_pitch;
        LELE_TRY(lele::pitched_out(out, pv, g.n, (int64_t)g.oc * g.plane, 4, (void**)&dst));
        g.obs = pv->out_pitch;
    } else {
        LELE_TRY(out->reserve((size_t)g.n * g.oc * g.plane * 4));
        dst = (float*)out->data;
    }
    LELE_TRY(run_conv2d(ctx, w, (const float*)dx, (const float*)dwp, (const float*)db, g, act, dst));
    return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, (int64_t)g.oh, (int64_t)g.ow});
}

int lele_hip_conv2d_res(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias, const LeleTensor* res,
                        const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                        const int64_t* strides, size_t nstr, int act, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape,
                        int32_t* out_rank) {
    LELE_REQUIRE(res, "conv2d_res: NULL residual");
    return conv2d_entry(ctx, x, w, bias, dilations, ndil, group, pads, npads, strides, nstr, act, pitch, out, out_shape, out_rank, res);
}

int lele_hip_conv2d(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                    const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                    const int64_t* strides, size_t nstr, int act, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    return conv2d_entry(ctx, x, w, bias, dilations, ndil, group, pads, npads, strides, nstr, act, nullptr, out, out_shape, out_rank);
}

int lele_hip_conv2d_pitched(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                            const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                            const int64_t* strides, size_t nstr, int act, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape,
                            int32_t* out_rank) {
    LELE_REQUIRE(pitch, "conv2d_pitched: pitch is NULL");
    return conv2d_entry(ctx, x, w, bias, dilations, ndil, group, pads, npads, strides, nstr, act, pitch, out, out_shape, out_rank);
}

/* reset_conv_stats / print_conv_stats (conv2d.rs:75,101; examples/yolo26n-seg/src/main.rs:64,74): counters of the 2-D convolutions
 * issued on this context since the last reset -- calls and multiply-accumulates (host arithmetic at issue: no device cost) */
int lele_hip_conv_stats_reset(LeleCtx* ctx) {
    LELE_REQUIRE(ctx, "conv_stats_reset: ctx is NULL");
    ctx->conv_calls = 0;
    ctx->conv_macs = 0;
    return 0;
}
int lele_hip_conv_stats(LeleCtx* ctx, int64_t* calls, int64_t* macs) {
    LELE_REQUIRE(ctx && calls && macs, "conv_stats: NULL argument");
    *calls = ctx->conv_calls;
    *macs = ctx->conv_macs;
    return 0;
}

int lele_hip_conv1d(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                    const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                    const int64_t* strides, size_t nstr, int relu, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && w && out, "conv1d: NULL argument");
    LELE_REQUIRE(x->rank == 3 || x->rank == 2, "Conv1d: Unsupported input rank %d", x->rank);  // conv1d.rs:867-873
    LELE_REQUIRE(w->rank == 3, "conv1d: expected weight [C_out,C_in/g,K]");
    LELE_REQUIRE(x->dtype == LELE_F32 && w->dtype == LELE_F32, "conv1d: f32 tensors required");
    LELE_REQUIRE(group >= 1, "conv1d: group must be >= 1");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    ConvGeom g{};
    g.n = (int)x->shape[0];
    g.c = x->rank == 3 ? (int)x->shape[1] : 1;  // [N, L] is a single channel
    g.ih = 1;
    g.iw = (int)x->shape[x->rank - 1];
    g.oc = (int)w->shape[0];
    g.kh = 1;
    g.kw = (int)w->shape[2];
    g.group = (int)group;
    LELE_REQUIRE(g.c % g.group == 0 && g.oc % g.group == 0, "conv1d: channels not divisible by group");
    g.icg = g.c / g.group;
    g.ocg = g.oc / g.group;
    LELE_REQUIRE(w->shape[1] == g.icg, "conv1d: weight C_in/g mismatch");
    g.dh = 1;
    g.dw = ndil ? (int)dilations[0] : 1;
    g.sh = 1;
    g.sw = nstr ? (int)strides[0] : 1;
    const int pl = npads >= 1 ? (int)pads[0] : 0, pr = npads >= 2 ? (int)pads[1] : 0;  // conv1d.rs:886-887
    g.pt = 0;
    g.pl = pl;
    const int64_t nw = (int64_t)g.iw + pl + pr - (int64_t)g.dw * (g.kw - 1) - 1;  // conv1d.rs:888-889
    LELE_REQUIRE(nw >= 0 && g.sw > 0, "conv1d: output length must be positive");
    g.oh = 1;
    g.ow = (int)(nw / g.sw + 1);
    g.K = g.icg * g.kw;
    g.plane = g.ow;
    if (bias) LELE_REQUIRE(numel(bias) >= g.oc, "conv1d: bias shorter than C_out");
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dwp = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(w, &dwp));
    if (bias) LELE_TRY(ctx->dev_ptr(bias, &db));
    LELE_TRY(out->reserve((size_t)g.n * g.oc * g.plane * 4));
    LELE_TRY(run_conv2d(ctx, w, (const float*)dx, (const float*)dwp, (const float*)db, g, relu ? LELE_ACT_RELU : LELE_ACT_NONE,
                        (float*)out->data));
    return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, (int64_t)g.ow});
}

int lele_hip_depthwise_conv1d_tlc(LeleCtx* ctx, const LeleTensor* x, int64_t x_offset, const LeleTensor* w, const LeleTensor* bias,
                                  int64_t pad_left, int64_t pad_right, int relu, int add_input, LeleBuf* out, int64_t* out_shape,
                                  int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && w && out, "depthwise_conv1d_tlc: NULL argument");
    LELE_REQUIRE(x->rank == 3 && w->rank == 3 && x->dtype == LELE_F32 && w->dtype == LELE_F32, "depthwise_conv1d_tlc: x [B,T,P] and w [C,1,K] f32 required");
    const int64_t bsz = x->shape[0], t_in = x->shape[1], pitch = x->shape[2], c = w->shape[0], k = w->shape[2];
    LELE_REQUIRE(w->shape[1] == 1, "depthwise_conv1d_tlc: weight must be [C, 1, K] (group = C)");
    LELE_REQUIRE(x_offset >= 0 && x_offset + c <= pitch, "depthwise_conv1d_tlc: channels [%lld, %lld) outside the last dimension (%lld)",
                 (long long)x_offset, (long long)(x_offset + c), (long long)pitch);
    LELE_REQUIRE(k == 3 || k == 5 || k == 7 || k == 11, "depthwise_conv1d_tlc: kernel sizes 3, 5, 7, 11 (use transpose + conv1d otherwise)");
    LELE_REQUIRE(pad_left >= 0 && pad_right >= 0, "depthwise_conv1d_tlc: negative padding");
    const int64_t t_out = t_in + pad_left + pad_right - (k - 1);
    LELE_REQUIRE(t_out >= 1 && t_in >= 1, "conv1d: output length must be positive");
    LELE_REQUIRE(!add_input || (t_out == t_in && pad_left <= k - 1), "depthwise_conv1d_tlc: the input can only be added to an output of the same length");
    if (bias) LELE_REQUIRE(numel(bias) >= c, "conv1d: bias shorter than C_out");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dwp = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(w, &dwp));
    if (bias) LELE_TRY(ctx->dev_ptr(bias, &db));
    LELE_TRY(out->reserve((size_t)(bsz * t_out * c) * 4));
    // time steps per thread: 8 (a window of 18 loads for 8 results); a grid that would not give every CU a workgroup that way -- one
    // utterance -- takes 4 (3.4 against 4.2 us at [504, 512]; at [32 x 171, 512]: 4 -> 12.4, 8 -> 10.2, 16 -> 10.0 us)
    const char* tt_env = lab_env("LELE_HIP_TLC_TT");
    const int tt = tt_env && *tt_env ? atoi(tt_env) : (bsz * ((t_out + 7) / 8) * c < 256 * (int64_t)ctx->num_cus ? 4 : 8);
    const int64_t tiles = (t_out + tt - 1) / tt, total = bsz * tiles * c;
    LELE_REQUIRE(total < (int64_t(1) << 31) && pitch < (int64_t(1) << 31), "depthwise_conv1d_tlc: tensor too large");
    if (bsz && c) {
        const dim3 grid((unsigned)((total + 255) / 256));
#define LELE_TLC2(KW, TT_)                                                                                                      \
    hipLaunchKernelGGL((dwconv1d_tlc_kernel<KW, TT_>), grid, dim3(256), 0, ctx->stream, (const float*)dx + x_offset, (const float*)dwp, \
                       (const float*)db, (float*)out->data, (int)t_in, (int)t_out, (int)c, (int)pitch, (int)pad_left, relu, add_input,  \
                       (unsigned)tiles, (unsigned)total)
#define LELE_TLC(KW)                      \
    do {                                  \
        if (tt == 16) LELE_TLC2(KW, 16);  \
        else if (tt <= 4) LELE_TLC2(KW, 4); \
        else LELE_TLC2(KW, 8);            \
    } while (0)
        if (k == 3) LELE_TLC(3);
        else if (k == 5) LELE_TLC(5);
        else if (k == 7) LELE_TLC(7);
        else LELE_TLC(11);
#undef LELE_TLC
#undef LELE_TLC2
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape(out_shape, out_rank, {bsz, t_out, c});
}

int lele_hip_conv_transpose(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* bias,
                            const int64_t* dilations, size_t ndil, int64_t group, const int64_t* pads, size_t npads,
                            const int64_t* strides, size_t nstr, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && w && out, "conv_transpose: NULL argument");
    LELE_REQUIRE(x->rank == 4, "ConvTranspose: expected rank-4 input [N,C,H,W], got rank %d", x->rank);
    LELE_REQUIRE(w->rank == 4, "ConvTranspose: expected rank-4 weight [C_in,C_out/g,kH,kW], got rank %d", w->rank);
    LELE_REQUIRE(group == 1, "ConvTranspose: group > 1 not supported yet");  // conv2d.rs:3042
    LELE_REQUIRE(w->shape[0] == x->shape[1], "ConvTranspose: weight C_in mismatch");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    CtGeom g{};
    g.n = (int)x->shape[0];
    g.c = (int)x->shape[1];
    g.ih = (int)x->shape[2];
    g.iw = (int)x->shape[3];
    g.oc = (int)w->shape[1];
    g.kh = (int)w->shape[2];
    g.kw = (int)w->shape[3];
    g.sh = nstr > 0 ? (int)strides[0] : 1;  // conv2d.rs:3007-3014
    g.sw = nstr > 1 ? (int)strides[1] : 1;
    g.dh = ndil > 0 ? (int)dilations[0] : 1;
    g.dw = ndil > 1 ? (int)dilations[1] : 1;
    g.pt = npads > 0 ? (int)pads[0] : 0;
    g.pl = npads > 1 ? (int)pads[1] : 0;
    const int pb = npads > 2 ? (int)pads[2] : g.pt, pr = npads > 3 ? (int)pads[3] : g.pl;
    const int64_t oh = (int64_t)(g.ih - 1) * g.sh - (g.pt + pb) + (int64_t)g.dh * (g.kh - 1) + 1;
    const int64_t ow = (int64_t)(g.iw - 1) * g.sw - (g.pl + pr) + (int64_t)g.dw * (g.kw - 1) + 1;
    LELE_REQUIRE(oh > 0 && ow > 0, "conv_transpose: output dimensions must be positive, got out_h=%lld out_w=%lld",
                 (long long)oh, (long long)ow);
    g.oh = (int)oh;
    g.ow = (int)ow;
    if (bias) LELE_REQUIRE(numel(bias) >= g.oc, "conv_transpose: bias shorter than C_out");
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dwp = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(w, &dwp));
    if (bias) LELE_TRY(ctx->dev_ptr(bias, &db));
    const int64_t total = (int64_t)g.n * g.oc * oh * ow;
    LELE_TRY(out->reserve((size_t)total * 4));
    if (total == 0) return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, oh, ow});
    if (lab_env("LELE_HIP_CONVT_GATHER")) {  // reference gather kernel, kept for A/B checks
        hipLaunchKernelGGL(conv_transpose_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float*)dx,
                           (const float*)dwp, (const float*)db, (float*)out->data, g);
        LELE_HIP_CHECK(hipGetLastError());
        return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, oh, ow});
    }
    if (g.kh == g.sh && g.kw == g.sw && g.dh == 1 && g.dw == 1 && g.pt == 0 && g.pl == 0 && pb == 0 && pr == 0 &&
        (int64_t)g.oc * g.kh * g.kw < (int64_t(1) << 31) && !lab_env("LELE_HIP_CONVT_PHASES")) {
        // kernel == stride: one GEMM with a scattering epilogue (see ConvTKsEpi)
        const int taps = g.kh * g.kw, mrows = g.oc * taps, plane_in = g.ih * g.iw;
        const size_t wb = (size_t)g.c * mrows * 4;
        void* dwk = nullptr;
        const bool cache_k = w->mem == LELE_MEM_WEIGHT;
        auto kkey = std::make_tuple((const void*)w->data, wb, 450);
        auto kit = cache_k ? ctx->weights.find(kkey) : ctx->weights.end();
        if (kit != ctx->weights.end()) {
            dwk = kit->second;
        } else {
            if (cache_k) {
                LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
                LELE_HIP_CHECK(hipMalloc(&dwk, std::max<size_t>(wb, 16)));
                ctx->weights[kkey] = dwk;
            } else {
                LELE_TRY(ctx->arena_alloc(std::max<size_t>(wb, 16), &dwk));
            }
            hipLaunchKernelGGL(convt_wks_kernel, dim3(grid_for((int64_t)g.c * mrows)), dim3(256), 0, ctx->stream, (const float*)dwp, (float*)dwk,
                               g.c, g.oc, taps);
        }
        ConvGeom q{};
        q.group = 1;
        q.ocg = mrows;
        q.K = g.c;
        ConvWLoad al{(const float*)dwk, q, (int)(g.c % 4 == 0)};
        gemm::LoadKRow bl{(const float*)dx, (int64_t)g.c * plane_in, (int64_t)plane_in, plane_in, g.c};
        ConvTKsEpi epi{(float*)out->data, (const float*)db, make_fastdiv(g.iw, plane_in), make_fastdiv(taps, mrows), make_fastdiv(g.kw, taps),
                       g.oc, g.oh, g.ow, g.kh, g.kw, g.iw, plane_in, taps};
        gemm::launch(ctx->stream, al, bl, epi, mrows, plane_in, g.c, g.n, ctx->num_cus);
        LELE_HIP_CHECK(hipGetLastError());
        return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, oh, ow});
    }
    // per-phase weights: one buffer holding every phase's [OC][K_phase] block (each tap belongs to exactly one phase)
    const size_t wbytes = (size_t)g.c * g.oc * g.kh * g.kw * 4;
    const int tap_major = g.c % 4 == 0;
    void* dwt = nullptr;
    const bool cacheable = w->mem == LELE_MEM_WEIGHT;
    // the phase structure depends on stride / dilation / pad-begin, so they are part of the cache tag
    const int tag = 400 + ((g.sh * 31 + g.sw) * 31 + g.dh) * 31 + g.dw + 7919 * (g.pt * 64 + g.pl);
    auto key = std::make_tuple((const void*)w->data, wbytes, tag);
    auto it = cacheable ? ctx->weights.find(key) : ctx->weights.end();
    const bool have_w = it != ctx->weights.end();
    if (have_w) {
        dwt = it->second;
    } else if (cacheable) {
        LELE_REQUIRE(!ctx->capturing, "graph capture: this op must run once eagerly first (it allocates or synchronises)");
        LELE_HIP_CHECK(hipMalloc(&dwt, std::max<size_t>(wbytes, 16)));
        ctx->weights[key] = dwt;
    } else {
        LELE_TRY(ctx->arena_alloc(std::max<size_t>(wbytes, 16), &dwt));
    }
    auto mod = [](int v, int m) { return ((v % m) + m) % m; };
    size_t woff = 0;
    for (int py = 0; py < g.sh; ++py)
        for (int px = 0; px < g.sw; ++px) {
            const int nj = py < g.oh ? (g.oh - py + g.sh - 1) / g.sh : 0, ni = px < g.ow ? (g.ow - px + g.sw - 1) / g.sw : 0;
            int a0 = -1, sa = 1, na = 0, b0 = -1, sb = 1, nb = 0;
            for (int a = 0; a < g.kh; ++a)
                if (mod(py + g.pt - a * g.dh, g.sh) == 0) {
                    if (na == 0) a0 = a;
                    if (na == 1) sa = a - a0;
                    ++na;
                }
            for (int b = 0; b < g.kw; ++b)
                if (mod(px + g.pl - b * g.dw, g.sw) == 0) {
                    if (nb == 0) b0 = b;
                    if (nb == 1) sb = b - b0;
                    ++nb;
                }
            const int kp = g.c * na * nb;
            float* wph = (float*)dwt + woff;
            woff += (size_t)g.oc * kp;
            if (nj == 0 || ni == 0) continue;
            if (kp == 0) {  // no tap reaches this phase: bias only
                hipLaunchKernelGGL(convt_fill_phase_kernel, dim3(grid_for((int64_t)g.n * g.oc * nj * ni)), dim3(256), 0,
                                   ctx->stream, (float*)out->data, (const float*)db, g.n, g.oc, g.oh, g.ow, py, px, g.sh,
                                   g.sw, nj, ni);
                continue;
            }
            if (!have_w)
                hipLaunchKernelGGL(convt_wphase_kernel, dim3(grid_for((int64_t)g.oc * kp)), dim3(256), 0, ctx->stream,
                                   (const float*)dwp, wph, g.c, g.oc, g.kh, g.kw, a0, sa, na, b0, sb, nb, tap_major);
            ConvGeom q{};
            q.n = g.n;
            q.c = g.c;
            q.ih = g.ih;
            q.iw = g.iw;
            q.oc = g.oc;
            q.kh = na;
            q.kw = nb;
            q.group = 1;
            q.icg = g.c;
            q.ocg = g.oc;
            q.sh = q.sw = 1;
            q.pt = -((py + g.pt - a0 * g.dh) / g.sh);  // iy = j - pt + t*dh with dh = -(sa*dh/sh)
            q.pl = -((px + g.pl - b0 * g.dw) / g.sw);
            q.dh = -(sa * g.dh / g.sh);
            q.dw = -(sb * g.dw / g.sw);
            q.oh = nj;
            q.ow = ni;
            q.K = kp;
            q.plane = nj * ni;
            q.xbs = (long long)g.c * g.ih * g.iw;  // the loaders address images through it (run_conv2d fills it in; this path builds its own)
            ConvWLoad al{wph, q, (int)((((uintptr_t)wph & 15) == 0) && kp % 4 == 0)};
            ConvTEpi epi{(float*)out->data, (const float*)db, make_fastdiv(ni, q.plane), g.oc, g.oh, g.ow, py, px, g.sh, g.sw, ni,
                         q.plane};
            if (tap_major) {
                ConvXLoadTap bl{(const float*)dx, q, make_fastdiv(ni, q.plane), make_fastdiv(g.c, kp), make_fastdiv(nb, na * nb)};
                gemm::launch(ctx->stream, al, bl, epi, g.oc, q.plane, kp, g.n, ctx->num_cus);
            } else {
                ConvXLoad bl{(const float*)dx, q, make_fastdiv(ni, q.plane), make_fastdiv(na * nb, kp), make_fastdiv(nb, na * nb)};
                gemm::launch(ctx->stream, al, bl, epi, g.oc, q.plane, kp, g.n, ctx->num_cus);
            }
        }
    LELE_HIP_CHECK(hipGetLastError());
    return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, oh, ow});
}

int lele_hip_conv_integer(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* w, const LeleTensor* x_zero_point,
                          const LeleTensor* w_zero_point, const int64_t* dilations, size_t ndil, int64_t group,
                          const int64_t* pads, size_t npads, const int64_t* strides, size_t nstr, LeleBuf* out,
                          int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && w && out, "conv_integer: NULL argument");
    LELE_REQUIRE(x->rank == 4 && w->rank == 4 && x->dtype == LELE_F32 && w->dtype == LELE_F32,
                 "conv_integer: rank-4 tensors holding u8 values as f32 required");  // conv2d.rs:1522-1523
    LELE_REQUIRE(!(x_zero_point && x_zero_point->mem == LELE_MEM_DEVICE) && !(w_zero_point && w_zero_point->mem == LELE_MEM_DEVICE),
                 "conv_integer: zero points are scalar attributes and must be host tensors");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    ConvGeom g;
    LELE_TRY(conv_geom_from(x, x->shape, w, dilations, ndil, group, pads, npads, strides, nstr, &g));
    const float x_zp = host_scalar(x_zero_point), w_zp = host_scalar(w_zero_point);
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)g.n * g.oc * g.plane * 4));
    const float* src = (const float*)dx;
    const int64_t src_c = g.c;
    LELE_TRY(ci_run(ctx, g, w, w_zp, &src, &src_c, 1, nullptr, x_zp, (float*)out->data));
    return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, (int64_t)g.oh, (int64_t)g.ow});
}

// conv_integer_from_f32 (conv2d.rs:2246) and conv_integer_from_f32_multi (:2420: channel-concatenated sources, 1x1 conv).
// nsrc == 1 with explicit attributes is the single form; out_scale receives the DynamicQuantizeLinear scale ([1], device).
int lele_hip_conv_integer_from_f32(LeleCtx* ctx, const LeleTensor* const* sources, size_t nsrc, const LeleTensor* w,
                                   const LeleTensor* w_zero_point, const int64_t* dilations, size_t ndil, int64_t group,
                                   const int64_t* pads, size_t npads, const int64_t* strides, size_t nstr, LeleBuf* out,
                                   LeleBuf* out_scale, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && sources && nsrc >= 1 && w && out && out_scale, "conv_integer_from_f32: NULL argument");
    LELE_REQUIRE(w->rank == 4 && w->dtype == LELE_F32, "conv_integer_from_f32: rank-4 f32-coded weights required");
    LELE_REQUIRE(!(w_zero_point && w_zero_point->mem == LELE_MEM_DEVICE), "conv_integer_from_f32: w_zero_point must be a host scalar");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t total_c = 0;
    for (size_t i = 0; i < nsrc; ++i) {
        const LeleTensor* s = sources[i];
        LELE_REQUIRE(s && s->rank == 4 && s->dtype == LELE_F32, "conv_integer_from_f32: sources must be rank-4 f32");
        LELE_REQUIRE(s->shape[0] == sources[0]->shape[0] && s->shape[2] == sources[0]->shape[2] && s->shape[3] == sources[0]->shape[3],
                     "conv_integer_from_f32_multi: sources must share N, H, W");
        total_c += s->shape[1];
    }
    const int64_t xshape[4] = {sources[0]->shape[0], total_c, sources[0]->shape[2], sources[0]->shape[3]};
    ConvGeom g;
    LELE_TRY(conv_geom_from(sources[0], xshape, w, dilations, ndil, group, pads, npads, strides, nstr, &g));
    const float w_zp = host_scalar(w_zero_point);
    LELE_TRY(ctx->arena_reset());
    std::vector<const float*> ds(nsrc);
    std::vector<int64_t> lens(nsrc);
    for (size_t i = 0; i < nsrc; ++i) {
        const void* d = nullptr;
        LELE_TRY(ctx->dev_ptr(sources[i], &d));
        ds[i] = (const float*)d;
        lens[i] = numel(sources[i]);
    }
    void* prm = nullptr;
    LELE_TRY(ctx->arena_alloc(sizeof(QParamsDev), &prm));
    LELE_TRY(out->reserve((size_t)g.n * g.oc * g.plane * 4));
    LELE_TRY(out_scale->reserve(4));
    LELE_TRY(quant_params_of(ctx, ds.data(), lens.data(), (int)nsrc, prm));
    hipLaunchKernelGGL(ci_scale_out_kernel, dim3(1), dim3(1), 0, ctx->stream, (const QParamsDev*)prm, (float*)out_scale->data);
    std::vector<int64_t> chans(nsrc);
    for (size_t i = 0; i < nsrc; ++i) chans[i] = sources[i]->shape[1];
    LELE_TRY(ci_run(ctx, g, w, w_zp, ds.data(), chans.data(), (int)nsrc, (const QParamsDev*)prm, 0.0f, (float*)out->data));
    return set_shape(out_shape, out_rank, {(int64_t)g.n, (int64_t)g.oc, (int64_t)g.oh, (int64_t)g.ow});
}

int lele_hip_fused_scale_bias(LeleCtx* ctx, const LeleTensor* data, const LeleTensor* scale_dev_or_null, float scale_mul,
                              const LeleTensor* bias, int silu, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && data && bias && out, "fused_scale_bias: NULL argument");
    LELE_REQUIRE(data->rank == 4 && data->dtype == LELE_F32, "fused_scale_bias: [N,C,H,W] f32 required");
    LELE_REQUIRE(numel(bias) >= data->shape[1], "fused_scale_bias: bias shorter than C");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(ctx->arena_reset());
    const void *dd = nullptr, *db = nullptr, *dsc = nullptr;
    LELE_TRY(ctx->dev_ptr(data, &dd));
    LELE_TRY(ctx->dev_ptr(bias, &db));
    if (scale_dev_or_null) LELE_TRY(ctx->dev_ptr(scale_dev_or_null, &dsc));
    const int64_t total = numel(data);
    LELE_TRY(out->reserve((size_t)total * 4));
    if (total) {
        hipLaunchKernelGGL(ci_scale_bias_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float*)dd, total,
                           data->shape[1], data->shape[2] * data->shape[3], (const float*)dsc, scale_mul, (const float*)db, silu,
                           (float*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(data->shape, data->shape + 4));
}

}  // extern "C"
