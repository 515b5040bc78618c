// manip.hip -- the data-movement operators (bit-exact class): pure index arithmetic on the device.
//
//   strided gather-copy engine  -> slice (manipulation.rs:209-380), transpose (644-1080), expand (math.rs:2168-2247),
//                                  tile (math.rs:2249-2302), concat (manipulation.rs:108-207), split/split_owned
//                                  (1091-1213)
//   pad (constant / edge / reflect)            manipulation.rs:382-587
//   gather                                      manipulation.rs:589-641
//   gather_elements, topk, resize_nearest, max_pool2d   conv2d.rs:1051-1502
//   range / constant_of_shape / cast            math.rs:2033-2082, shape.rs:122-135, utils.rs:66-101
// Views (reshape, flatten, squeeze, unsqueeze, identity; shape.rs:2-186) move no data: the host mirror handles them.
// Elements are moved as 4- or 8-byte words, so every dtype of TensorView (f32, i32, i64) goes through the same code.
#include "common.h"

#include <math.h>

using namespace lele;

namespace {

struct CopyDesc {
    int rank;
    int64_t oshape[LELE_MAX_RANK];
    int64_t istride[LELE_MAX_RANK];  // element strides into the source (may be 0 or negative)
    int64_t imod[LELE_MAX_RANK];     // >0: source coordinate = coordinate % imod (tile)
    int64_t ostride[LELE_MAX_RANK];  // element strides into the destination
    int64_t ioff, ooff;
};

// I = int32_t when every index fits (the common case: 32-bit div/mod), int64_t otherwise
template <typename W, typename I>
__global__ void strided_copy_kernel(const W* __restrict__ in, W* __restrict__ out, int64_t numel, CopyDesc d) {
    for (int64_t i0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < numel; i0 += (int64_t)gridDim.x * blockDim.x) {
        I rem = (I)i0, si = (I)d.ioff, di = (I)d.ooff;
        for (int k = d.rank - 1; k >= 0; --k) {
            const I sh = (I)d.oshape[k];
            I c = rem % sh;
            rem /= sh;
            di += c * (I)d.ostride[k];
            if (d.imod[k] > 0) c %= (I)d.imod[k];
            si += c * (I)d.istride[k];
        }
        out[di] = in[si];
    }
}

// Transposing copies: the innermost output dim is strided in the input while some other dim `tj` is contiguous there.
// A 64x64 tile goes through LDS so that both the reads (along tj) and the writes (along the inner dim) are coalesced
// 256-byte rows.  Every thread issues its 16 loads back to back from CLAMPED coordinates (a load behind a per-lane bounds
// branch costs one serialised memory round trip each); only the stores are predicated.
// grid.x = tiles along the inner dim, grid.y = tiles along tj, grid.z = all remaining dims flattened.
template <typename W>
__global__ __launch_bounds__(256) void transpose_tile_kernel(const W* __restrict__ in, W* __restrict__ out, CopyDesc d, int tj) {
    __shared__ W tile[64][65];
    const int r = d.rank, tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    int64_t rem = blockIdx.z, si = d.ioff, di = d.ooff;
    for (int k = r - 2; k >= 0; --k) {
        if (k == tj) continue;
        const int64_t c = rem % d.oshape[k];
        rem /= d.oshape[k];
        si += c * d.istride[k];
        di += c * d.ostride[k];
    }
    const int64_t j0 = (int64_t)blockIdx.y * 64, i0 = (int64_t)blockIdx.x * 64;
    const int64_t nj = d.oshape[tj], ni = d.oshape[r - 1];
    // read: tx runs along tj (input-contiguous), ty (+4 per step) along the inner dim
    {
        const int64_t j = j0 + tx < nj ? j0 + tx : nj - 1;
        const W* src = in + si + j * d.istride[tj];
        const int64_t is = d.istride[r - 1];
        W v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t i = i0 + ty + 4 * u;
            v[u] = src[(i < ni ? i : ni - 1) * is];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) tile[ty + 4 * u][tx] = v[u];
    }
    __syncthreads();
    // write: tx runs along the inner dim (output-contiguous), ty along tj
    {
        const int64_t i = i0 + tx;
        W* dst = out + di + i * d.ostride[r - 1];
        const int64_t os = d.ostride[tj];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t j = j0 + ty + 4 * u;
            if (j < nj && i < ni) dst[j * os] = tile[tx][ty + 4 * u];
        }
    }
}

template <typename W>
__global__ void fill_kernel(W* __restrict__ out, int64_t numel, W v) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = v;
}

struct PadDesc {
    int rank, mode;  // 0 constant, 1 edge, 2 reflect
    int64_t oshape[LELE_MAX_RANK], ishape[LELE_MAX_RANK], before[LELE_MAX_RANK];
};
template <typename W>
__global__ void pad_kernel(const W* __restrict__ in, W* __restrict__ out, int64_t numel, PadDesc d, W fill) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = i, si = 0, st = 1;
        bool inside = true;
        for (int k = d.rank - 1; k >= 0; --k) {
            const int64_t c = rem % d.oshape[k];
            rem /= d.oshape[k];
            int64_t s = c - d.before[k];
            if (s < 0 || s >= d.ishape[k]) {
                inside = false;
                if (d.mode == 1)
                    s = s < 0 ? 0 : d.ishape[k] - 1;  // manipulation.rs:511-518
                else if (d.mode == 2)
                    // manipulation.rs:562-569 AS WRITTEN: mirrored without the edge at the front (pad_begin +
                    // (pad_begin - c)) but WITH the edge at the back (pad_end - 1 - past); kept for bit-parity
                    s = s < 0 ? -s : 2 * d.ishape[k] - 1 - s;
            }
            si += s * st;
            st *= d.ishape[k];
        }
        out[i] = (inside || d.mode != 0) ? in[si] : fill;
    }
}

// gather (manipulation.rs:589-641): out[o][j][k] = data[o][idx[j]][k]
template <typename W, typename I>
__global__ void gather_kernel(const W* __restrict__ data, const I* __restrict__ idx, W* __restrict__ out,
                              int64_t outer, int64_t axis_dim, int64_t inner, int64_t nidx, unsigned* __restrict__ deverr) {
    const int64_t total = outer * nidx * inner;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i % inner, j = (i / inner) % nidx, o = i / (inner * nidx);
        int64_t v = (int64_t)idx[j];
        if (v < 0) v += axis_dim;
        // the reference indexes a slice here and panics when out of range (manipulation.rs:626-633): clamp the read and raise
        // the ctx's sticky error word (host-mapped; reported at the next sync / buf_to_host)
        if (v < 0 || v >= axis_dim) {
            *deverr = LELE_DEVERR_GATHER_INDEX;
            v = v < 0 ? 0 : axis_dim - 1;
        }
        out[i] = data[(o * axis_dim + v) * inner + k];
    }
}

// gather_elements (conv2d.rs:1438-1502): indices carried as f32
struct GeDesc {
    int rank, axis;
    int64_t ishape[LELE_MAX_RANK], xstride[LELE_MAX_RANK], axis_dim;
};
__global__ void gather_elements_kernel(const float* __restrict__ x, const float* __restrict__ idx,
                                       float* __restrict__ out, int64_t total, GeDesc d, unsigned* __restrict__ deverr) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t rem = i, off = 0;
        int64_t iv = (int64_t)idx[i];
        if (iv < 0) iv += d.axis_dim;
        if (iv < 0 || iv >= d.axis_dim) {  // conv2d.rs:1480-1490 indexes the input slice: a panic upstream
            *deverr = LELE_DEVERR_GATHER_INDEX;
            iv = iv < 0 ? 0 : d.axis_dim - 1;
        }
        for (int k = d.rank - 1; k >= 0; --k) {
            const int64_t c = rem % d.ishape[k];
            rem /= d.ishape[k];
            off += (k == d.axis ? iv : c) * d.xstride[k];
        }
        out[i] = x[off];
    }
}

// adaptive_avg_pool1d (pooling.rs:1-30): out[c][i] = sum(in[c][start..end]) / (end - start) with start = floor(i*L/O),
// end = ceil((i+1)*L/O), both clamped to L; an empty window yields 0.  The window is summed in index order (as upstream).
__global__ void adaptive_avg_pool1d_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t channels, int64_t in_len,
                                           int64_t out_len) {
    const int64_t total = channels * out_len;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = t / out_len, i = t - c * out_len;
        int64_t start = (i * in_len) / out_len, end = ((i + 1) * in_len + out_len - 1) / out_len;
        start = start < in_len ? start : in_len;
        end = end < in_len ? end : in_len;
        const int64_t len = end - start;
        float sum = 0.0f;
        const float* p = x + c * in_len;
        for (int64_t k = start; k < end; ++k) sum = sum + p[k];
        out[t] = len == 0 ? 0.0f : sum / (float)len;
    }
}

// resize_nearest (conv2d.rs:1261-1382): f32 coordinate arithmetic exactly as the reference
__global__ void resize_nearest_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t planes, int in_h,
                                      int in_w, int out_h, int out_w, int asymmetric, int channels, long long xbs, long long obs) {
    const float h_scale = (float)in_h / (float)out_h, w_scale = (float)in_w / (float)out_w;
    const int64_t total = planes * out_h * out_w;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int ow = (int)(i % out_w), oh = (int)((i / out_w) % out_h);
        const int64_t p = i / ((int64_t)out_w * out_h);
        int ih, iw;
        if (asymmetric) {
            ih = (int)fminf(floorf((float)oh * h_scale), (float)(in_h - 1));
            iw = (int)fminf(floorf((float)ow * w_scale), (float)(in_w - 1));
        } else {
            ih = (int)fminf(fmaxf(roundf(((float)oh + 0.5f) * h_scale - 0.5f), 0.0f), (float)(in_h - 1));
            iw = (int)fminf(fmaxf(roundf(((float)ow + 0.5f) * w_scale - 0.5f), 0.0f), (float)(in_w - 1));
        }
        // plane p = (image, channel); xbs / obs = elements from one image to the next (dense, or a channel view: LelePitch)
        const int64_t img = p / channels, ch = p - img * channels;
        out[img * obs + (ch * out_h + oh) * (int64_t)out_w + ow] = x[img * xbs + (ch * in_h + ih) * (int64_t)in_w + iw];
    }
}

// asymmetric up-sampling by a power of two S (the x2 of a detection neck, the x4 / x8 of a prototype branch):
// out[S y + a][S x + b] = in[y][x] -- with h_scale = w_scale = 1 / S exactly representable, the reference's floor(o * scale) is
// o / S exactly.  A thread reads FOUR consecutive inputs (16 bytes) and writes S rows of 4 S outputs as 16-byte stores, whole rows of
// a wave contiguous on both sides.  The general kernel above writes 4 bytes per thread from an input index it derives with two
// divisions: 0.17 of the HBM rate on [64, 128, 80, 80].
template <int S>
__global__ __launch_bounds__(256) void resize_up_kernel(const float* __restrict__ x, float* __restrict__ out, unsigned quads_per_image,
                                                        int in_w4 /* in_w / 4 */, long long xbs, long long obs) {
    const unsigned q = blockIdx.x * 256u + threadIdx.x;
    if (q >= quads_per_image) return;
    const unsigned row = q / (unsigned)in_w4, xq = q - row * (unsigned)in_w4;   // row = channel * in_h + y
    const float4 v = *reinterpret_cast<const float4*>(x + (long long)blockIdx.y * xbs + (long long)row * (4 * in_w4) + 4 * xq);
    const float e[4] = {v.x, v.y, v.z, v.w};
    const long long out_w = (long long)S * 4 * in_w4;
    float* o = out + (long long)blockIdx.y * obs + (long long)row * S * out_w + (long long)S * 4 * xq;
    float4 seg[S];
#pragma unroll
    for (int j = 0; j < S; ++j) seg[j] = make_float4(e[(4 * j) / S], e[(4 * j + 1) / S], e[(4 * j + 2) / S], e[(4 * j + 3) / S]);
#pragma unroll
    for (int a = 0; a < S; ++a)
#pragma unroll
        for (int j = 0; j < S; ++j) reinterpret_cast<float4*>(o + a * out_w)[j] = seg[j];
}

// max_pool2d (conv2d.rs:1051-1254): padded cells are skipped (== -inf)
struct PoolDesc {
    int in_h, in_w, out_h, out_w, kh, kw, sh, sw, pt, pl, dh, dw;
    int channels;        // planes per image
    long long xbs, obs;  // elements from one image to the next in x / out (dense, or a channel view: LelePitch)
};
// one thread per output; 32-bit index arithmetic when the tensor allows it; every tap is an unconditional load from a
// clamped address followed by a select (a bounds-checked load serialises one memory round trip per tap)
template <typename I>
__global__ __launch_bounds__(256) void max_pool2d_kernel(const float* __restrict__ x, float* __restrict__ out, int64_t planes,
                                                         PoolDesc d) {
    const I plane_out = (I)d.out_h * d.out_w, plane_in = (I)d.in_h * d.in_w;
    const I total = (I)planes * plane_out;
    for (I i = (I)blockIdx.x * 256 + threadIdx.x; i < total; i += (I)gridDim.x * 256) {
        const I p = i / plane_out;
        const int r = (int)(i - p * plane_out);
        const int oh = r / d.out_w, ow = r - oh * d.out_w;
        const I img = p / (I)d.channels, ch = p - img * (I)d.channels;
        const float* xp = x + (int64_t)img * d.xbs + (int64_t)ch * plane_in;
        const int ih0 = oh * d.sh - d.pt, iw0 = ow * d.sw - d.pl;
        float m = -INFINITY;
        for (int a = 0; a < d.kh; ++a) {
            const int ih = ih0 + a * d.dh;
            const bool hin = ih >= 0 && ih < d.in_h;
            const int rowoff = (hin ? ih : 0) * d.in_w;
#pragma unroll 5
            for (int b = 0; b < d.kw; ++b) {
                const int iw = iw0 + b * d.dw;
                const bool in = hin && iw >= 0 && iw < d.in_w;
                const float v = xp[rowoff + (in ? iw : 0)];
                m = (in && v > m) ? v : m;
            }
        }
        out[(int64_t)img * d.obs + (int64_t)ch * plane_out + r] = m;
    }
}

// The same pooling for planes that fit in LDS (the 5 x 5 / stride 1 pools of an SPPF block run on 20 x 20 maps): a workgroup
// copies PB whole input planes of ONE image -- a contiguous run of global memory, 16-byte loads -- into LDS, every thread scans its
// windows there in the reference's (kh, kw) order with the reference's comparison (a padded cell or a NaN never wins, the first of
// equal values stays: conv2d.rs:1051-1254), and the results leave as one contiguous run.  The one-output-per-thread kernel above
// reads every input 25 times through the vector cache with a 4-byte lane stride: 0.1 of the HBM rate on [64, 128, 20, 20].
// grid (ceil(channels / PB), images); dynamic LDS = PB * (plane_in padded to 4 + plane_out padded to 4) floats.
__global__ __launch_bounds__(256) void max_pool2d_lds_kernel(const float* __restrict__ x, float* __restrict__ out, PoolDesc d, int pb, int separable) {
    extern __shared__ __attribute__((aligned(16))) float mp_lds[];
    const unsigned plane_in = (unsigned)(d.in_h * d.in_w), plane_out = (unsigned)(d.out_h * d.out_w);
    const unsigned c0 = blockIdx.x * (unsigned)pb, np = (unsigned)d.channels - c0 < (unsigned)pb ? (unsigned)d.channels - c0 : (unsigned)pb;
    const float* src = x + (long long)blockIdx.y * d.xbs + (long long)c0 * plane_in;
    float* dst = out + (long long)blockIdx.y * d.obs + (long long)c0 * plane_out;
    float* lin = mp_lds;
    float* lout = mp_lds + (((unsigned)pb * plane_in + 3u) & ~3u);
    float* lrow = separable ? lout + (unsigned)pb * plane_out : nullptr;  // [planes][in_h][out_w]: the row maxima
    {
        const unsigned count = np * plane_in;
        if ((((uintptr_t)src) & 15) == 0) {
            const unsigned nv = count >> 2;
            for (unsigned e = threadIdx.x; e < nv; e += 256u) reinterpret_cast<float4*>(lin)[e] = reinterpret_cast<const float4*>(src)[e];
            for (unsigned e = 4 * nv + threadIdx.x; e < count; e += 256u) lin[e] = src[e];
        } else {
            for (unsigned e = threadIdx.x; e < count; e += 256u) lin[e] = src[e];
        }
    }
    __syncthreads();
    if (lrow) {
        // SEPARABLE: the maximum of each input row's window first, then the maximum over the window's rows.  The value the (kh, kw)
        // scan keeps is the first element, in row-major order, that nothing later exceeds: the first row holding a maximal element,
        // and the first such element in it -- which is what a first-wins row pass followed by a first-wins column pass selects (a
        // NaN never wins either pass; only +0 / -0 can tell).  kh + kw LDS reads per output instead of kh x kw.
        const unsigned rows_w = (unsigned)d.in_h * (unsigned)d.out_w;
        for (unsigned it = threadIdx.x; it < np * rows_w; it += 256u) {
            const unsigned p = it / rows_w, r = it - p * rows_w;
            const int y = (int)(r / (unsigned)d.out_w), ow = (int)(r - (unsigned)y * (unsigned)d.out_w);
            const float* xr = lin + p * plane_in + y * d.in_w;
            const int iw0 = ow * d.sw - d.pl;
            float m = -INFINITY;
#pragma unroll 5
            for (int b = 0; b < d.kw; ++b) {
                const int iw = iw0 + b * d.dw;
                const bool in = iw >= 0 && iw < d.in_w;
                const float v = xr[in ? iw : 0];
                m = (in && v > m) ? v : m;
            }
            lrow[it] = m;
        }
        __syncthreads();
        for (unsigned it = threadIdx.x; it < np * plane_out; it += 256u) {
            const unsigned p = it / plane_out, r = it - p * plane_out;
            const int oh = (int)(r / (unsigned)d.out_w), ow = (int)(r - (unsigned)oh * (unsigned)d.out_w);
            const float* xc = lrow + p * rows_w + ow;
            const int ih0 = oh * d.sh - d.pt;
            float m = -INFINITY;
#pragma unroll 5
            for (int a = 0; a < d.kh; ++a) {
                const int ih = ih0 + a * d.dh;
                const bool in = ih >= 0 && ih < d.in_h;
                const float v = xc[(in ? ih : 0) * d.out_w];
                m = (in && v > m) ? v : m;
            }
            lout[it] = m;
        }
    } else
    for (unsigned it = threadIdx.x; it < np * plane_out; it += 256u) {
        const unsigned p = it / plane_out, r = it - p * plane_out;
        const int oh = (int)(r / (unsigned)d.out_w), ow = (int)(r - (unsigned)oh * (unsigned)d.out_w);
        const float* xp = lin + p * plane_in;
        const int ih0 = oh * d.sh - d.pt, iw0 = ow * d.sw - d.pl;
        float m = -INFINITY;
        for (int a = 0; a < d.kh; ++a) {
            const int ih = ih0 + a * d.dh;
            const bool hin = ih >= 0 && ih < d.in_h;
            const int rowoff = (hin ? ih : 0) * d.in_w;
#pragma unroll 5
            for (int b = 0; b < d.kw; ++b) {
                const int iw = iw0 + b * d.dw;
                const bool in = hin && iw >= 0 && iw < d.in_w;
                const float v = xp[rowoff + (in ? iw : 0)];
                m = (in && v > m) ? v : m;
            }
        }
        lout[it] = m;
    }
    __syncthreads();
    {
        const unsigned count = np * plane_out;
        if ((((uintptr_t)dst) & 15) == 0) {
            const unsigned nv = count >> 2;
            for (unsigned e = threadIdx.x; e < nv; e += 256u) reinterpret_cast<float4*>(dst)[e] = reinterpret_cast<const float4*>(lout)[e];
            for (unsigned e = 4 * nv + threadIdx.x; e < count; e += 256u) dst[e] = lout[e];
        } else {
            for (unsigned e = threadIdx.x; e < count; e += 256u) dst[e] = lout[e];
        }
    }
}

// topk over the last axis (conv2d.rs:1385-1435): stable sort by value => rank(i) = #{j : v_j beats v_i, or ties with
// j < i}.  One block per row, O(n^2 / threads) comparisons; rows of the sizes lele uses (<= 8400) take microseconds.
__global__ __launch_bounds__(256) void topk_kernel(const float* __restrict__ x, int64_t n, int64_t k, int largest,
                                                   float* __restrict__ values, float* __restrict__ indices) {
    // grid (ceil(n / 256), rows): every thread ranks ONE element by streaming the row through LDS in 2048-element chunks;
    // a workgroup stops as soon as all of its elements are known to be outside the top k.
    __shared__ __attribute__((aligned(16))) float chunk[2048];
    __shared__ int alive;
    const int64_t rowi = blockIdx.y;
    const float* row = x + rowi * n;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool in = i < n;
    const float v = in ? row[i] : 0.0f;
    int64_t rank = in ? 0 : k;
    for (int64_t c0 = 0; c0 < n; c0 += 2048) {
        __syncthreads();
        if (threadIdx.x == 0) alive = 0;
        const int64_t cn = n - c0 < 2048 ? n - c0 : 2048;
        const int nq = (int)((cn + 3) >> 2);  // float4 groups that hold anything: a row of 80 scans 20 of them, not 512
        for (int t = threadIdx.x; t < 4 * nq; t += 256) chunk[t] = t < cn ? row[c0 + t] : __builtin_nanf("");
        __syncthreads();
        if (rank < k) {
            // four candidates per LDS read; slots past cn hold NaN, which never counts.  A chunk that lies entirely
            // before element i counts "beats or ties" (ties with a lower index rank first), one entirely after counts
            // "beats" only; just the chunk containing i needs the per-candidate index test.  256 | 2048, so the case is
            // uniform across the workgroup.
            const int before = (int)(i - c0);
            int add = 0;
            const float4* c4 = reinterpret_cast<const float4*>(chunk);
            if (before >= 2048) {
#pragma unroll 8
                for (int t = 0; t < nq; ++t) {
                    const float4 u = c4[t];
                    if (largest)
                        add += (int)(u.x >= v) + (int)(u.y >= v) + (int)(u.z >= v) + (int)(u.w >= v);
                    else
                        add += (int)(u.x <= v) + (int)(u.y <= v) + (int)(u.z <= v) + (int)(u.w <= v);
                }
            } else if (before < 0) {
#pragma unroll 8
                for (int t = 0; t < nq; ++t) {
                    const float4 u = c4[t];
                    if (largest)
                        add += (int)(u.x > v) + (int)(u.y > v) + (int)(u.z > v) + (int)(u.w > v);
                    else
                        add += (int)(u.x < v) + (int)(u.y < v) + (int)(u.z < v) + (int)(u.w < v);
                }
            } else {
                for (int t = 0; t < nq; ++t) {
                    const float4 u = c4[t];
                    const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bool beats = largest ? (uu[e] > v) : (uu[e] < v);
                        add += (beats || (uu[e] == v && 4 * t + e < before)) ? 1 : 0;
                    }
                }
            }
            rank += add;
            if (rank < k) alive = 1;  // benign race: every writer stores 1
        }
        __syncthreads();
        if (!alive) break;
    }
    if (in && rank < k) {
        values[rowi * k + rank] = v;
        indices[rowi * k + rank] = (float)i;  // indices are returned as f32
    }
}

// Long rows (n > 2048): prefilter.  If t0 is the k-th best of the first 2048 elements, anything strictly worse than t0
// already has k elements ahead of it and cannot be in the top k -- and it can never outrank a survivor either, so the
// survivors' ranks depend on survivors only.  Three small kernels: threshold, compaction (value + original index, the
// index keeps the tie-break "lower index first"), rank among the survivors.
__global__ __launch_bounds__(256) void topk_threshold_kernel(const float* __restrict__ x, int64_t n, int64_t k, int largest,
                                                             float* __restrict__ t0, int* __restrict__ count) {
    // grid (32, rows): 2048 sample elements, FOUR lanes per element (each scans a quarter of the sample held in LDS; the four
    // partial ranks meet by shuffle) -- the serial 2048-step scan per element was the longest kernel of the YOLO post-processing
    __shared__ __attribute__((aligned(16))) float chunk[2048];
    const float* row = x + (int64_t)blockIdx.y * n;
    const int m = (int)(n < 2048 ? n : 2048);
    for (int t = threadIdx.x; t < 2048; t += 256) chunk[t] = t < m ? row[t] : __builtin_nanf("");
    __syncthreads();
    const int e = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const float v = chunk[e < 2048 ? e : 2047];
    int rank = 0;
    const float4* c4 = reinterpret_cast<const float4*>(chunk) + 128 * part;
#pragma unroll 4
    for (int t = 0; t < 128; ++t) {
        const float4 u = c4[t];
        const float uu[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const bool beats = largest ? (uu[q] > v) : (uu[q] < v);
            rank += (beats || (uu[q] == v && 512 * part + 4 * t + q < e)) ? 1 : 0;
        }
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    if (part == 0 && e < m && rank == k - 1) t0[blockIdx.y] = v;  // exactly one element of the sample has this rank
}
__global__ void topk_init_kernel(int64_t rows, int largest, float* __restrict__ t0, int* __restrict__ count) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < rows) {
        count[r] = 0;
        t0[r] = largest ? -INFINITY : INFINITY;  // stays when the sample holds fewer than k elements: keep everything
    }
}
__global__ __launch_bounds__(256) void topk_compact_kernel(const float* __restrict__ x, int64_t n, int largest,
                                                           const float* __restrict__ t0, int* __restrict__ count,
                                                           float* __restrict__ cv, int* __restrict__ ci) {
    const int64_t rowi = blockIdx.y, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[rowi * n + i], t = t0[rowi];
    const bool keep = largest ? !(v < t) : !(v > t);
    if (keep) {
        const int pos = atomicAdd(&count[rowi], 1);
        cv[rowi * n + pos] = v;
        ci[rowi * n + pos] = (int)i;
    }
}
__global__ __launch_bounds__(256) void topk_rank_kernel(const float* __restrict__ cv, const int* __restrict__ ci, int64_t n,
                                                        int64_t k, int largest, const int* __restrict__ count,
                                                        float* __restrict__ values, float* __restrict__ indices) {
    // 64 candidates per workgroup, four lanes each: lane `part` ranks its candidate against every fourth survivor of the
    // LDS chunk, the partial ranks meet by shuffle
    __shared__ float sv[1024];
    __shared__ int si[1024];
    const int64_t rowi = blockIdx.y;
    const int m = count[rowi];
    if ((int64_t)blockIdx.x * 64 >= m) return;  // uniform per workgroup
    const int j = blockIdx.x * 64 + (threadIdx.x >> 2), part = threadIdx.x & 3;
    const bool in = j < m;
    const float v = in ? cv[rowi * n + j] : 0.0f;
    const int vi = in ? ci[rowi * n + j] : 0;
    int rank = 0;
    for (int c0 = 0; c0 < m; c0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) {
            const bool live = c0 + t < m;
            sv[t] = live ? cv[rowi * n + c0 + t] : __builtin_nanf("");
            si[t] = live ? ci[rowi * n + c0 + t] : 0;
        }
        __syncthreads();
#pragma unroll 4
        for (int t = part; t < 1024; t += 4) {
            const float u = sv[t];
            const bool beats = largest ? (u > v) : (u < v);
            rank += (beats || (u == v && si[t] < vi)) ? 1 : 0;
        }
    }
    rank += __shfl_xor(rank, 1);
    rank += __shfl_xor(rank, 2);
    if (in && part == 0 && rank < k) {
        values[rowi * k + rank] = v;
        indices[rowi * k + rank] = (float)vi;
    }
}

// out[n][p][c] = x[n][c][p] for one image-pitched operand and result: 32 x 32 tiles through LDS, reads along p, writes along c
__global__ __launch_bounds__(256) void transpose_cp_kernel(const float* __restrict__ x, float* __restrict__ out, int c, int pos, long long xbs, long long obs) {
    __shared__ float t[32][33];
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const float* src = x + (long long)blockIdx.z * xbs;
    float* dst = out + (long long)blockIdx.z * obs;
    for (int i = threadIdx.x; i < 32 * 32; i += 256) {
        const int cl = i >> 5, pl = i & 31;
        if (c0 + cl < c && p0 + pl < pos) t[cl][pl] = src[(long long)(c0 + cl) * pos + p0 + pl];
    }
    __syncthreads();
    const int cw = c - c0 < 32 ? c - c0 : 32;
    for (int i = threadIdx.x; i < 32 * cw; i += 256) {
        const int pl = i / cw, cl = i - pl * cw;
        if (p0 + pl < pos) dst[(long long)(p0 + pl) * c + c0 + cl] = t[cl][pl];
    }
}

// Long rows, one workgroup per row: RADIX SELECT.  The prefilter above keeps everything at least as good as the k-th best of the
// first 2048 elements -- ~n k / 2048 survivors, ranked against each other in O(m^2): 0.51 ms for the [64, 24000] -> 300 selection
// of a Yolo26n-seg tail.  Here the k-th best value itself is found by four 8-bit histogram passes over order-preserving integer
// images of the values (the histogram bin that holds the k-th element fixes 8 more bits of it each pass), the elements better
// than it plus the first ties in index order are collected (exactly k of them: the stable sort's "lower index first" among equal
// values), and only those k are ranked.  Five sweeps of an L2-resident row and k^2 comparisons, no atomics on global memory, a
// deterministic result: the bits of topk_kernel.  (-0 counts as +0, as `==` does; a NaN ranks below everything.)
__device__ __forceinline__ unsigned topk_key(float v, int largest) {
    unsigned u = __float_as_uint(v);
    if (v != v) return 0u;
    if (v == 0.0f) u = 0u;
    const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending with the value; >= 1 for every non-NaN
    return largest ? asc : ~asc + 1u;                                   // "better" = larger key either way (~asc + 1 >= 1 too)
}
// (1024 threads a row; the bin search and the index-order scans are wave scans -- written with thread 0 walking the 256 bins and the
// per-thread counts one LDS read at a time, those serial walks were most of the kernel: 126 us for [64, 24000] -> 300)
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(v, off);
        if (lane >= off) v += o;
    }
    return v;
}
// STAGE (rows of at most TOPK_STAGE_MAX elements): the row's keys are fetched ONCE into LDS, many loads in flight, and the four
// histogram sweeps and the two collecting sweeps read LDS -- swept from L2 each time, a sweep was a chain of dependent round trips
// (one load, one vote, one atomic per trip): 75 us for [64, 24000] -> 300 and 47 us for [64, 8400] -> 300 on the Yolo26n-seg tail.
constexpr int TOPK_TPB = 1024, TOPK_STAGE_MAX = 28672;
template <bool STAGE>
__global__ __launch_bounds__(TOPK_TPB) void topk_select_kernel(const float* __restrict__ x, int64_t n64, int k, int largest,
                                                              float* __restrict__ values, float* __restrict__ indices) {
    extern __shared__ unsigned s_keys[];  // STAGE: [n]
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_need;
    __shared__ unsigned w_gt[TOPK_TPB / 64], w_eq[TOPK_TPB / 64];
    __shared__ unsigned sel_key[1024];
    __shared__ int sel_idx[1024];
    __shared__ float sel_val[1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = (int)n64;
    const float* row = x + (int64_t)blockIdx.x * n64;
    if (tid == 0) {
        s_prefix = 0u;
        s_need = (unsigned)k;
    }
    if (STAGE) {
        for (int i0 = 0; i0 < n; i0 += 8 * TOPK_TPB) {  // eight loads a thread in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TOPK_TPB + tid;
                v[u] = row[i < n ? i : n - 1];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u * TOPK_TPB + tid;
                if (i < n) s_keys[i] = topk_key(v[u], largest);
            }
        }
    }
    auto key_at = [&](int i) { return STAGE ? s_keys[i] : topk_key(row[i], largest); };
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix;
        for (int i0 = 0; i0 < n; i0 += TOPK_TPB) {  // (uniform trip count: the wave votes below need every lane)
            const int i = i0 + tid;
            const unsigned key = i < n ? key_at(i) : 0u;
            const bool act = i < n && (key & mask) == prefix;
            const unsigned bin = (key >> shift) & 255u;
            // Scores of one detection head sit close together: in the first passes (almost) every element of a wave falls into the SAME
            // bin, and 64 LDS atomics on one address serialise (the whole selection was 76 us for [64, 24000] -> 300 on such rows).  When
            // the live lanes of a wave agree on the bin, one of them adds the count.  (One count per DISTINCT bin -- a vote per bin, up
            // to eight, plain atomics for the rest -- measured slower on spread-out scores: 110 against 68 us.)
            const unsigned long long live = __ballot(act);
            if (live) {
                const unsigned first = (unsigned)__shfl((int)bin, (int)__builtin_ctzll(live));
                if (__ballot(act && bin != first) == 0ull) {
                    if (lane == (int)__builtin_ctzll(live)) atomicAdd(&hist[first], (unsigned)__popcll(live));
                } else if (act) {
                    atomicAdd(&hist[bin], 1u);
                }
            }
        }
        __syncthreads();
        if (wave == 0) {  // the bin (from the best downwards) in which the need-th element lies: lane l owns bins 4 (63 - l) + [0, 4)
            const int b0 = 4 * (63 - lane);
            const unsigned h0 = hist[b0], h1 = hist[b0 + 1], h2 = hist[b0 + 2], h3 = hist[b0 + 3];
            const unsigned mine = h0 + h1 + h2 + h3, upto = wave_incl_scan(mine, lane);  // elements in this lane's bins and all better ones
            const unsigned need = s_need, above = upto - mine;
            // the last lane takes whatever is left (bin 0 included), as the serial walk did
            if ((above < need && need <= upto) || (lane == 63 && need > upto)) {
                unsigned acc = above;
                int b = b0 + 3;
                const unsigned hb[4] = {h0, h1, h2, h3};
                for (; b > b0; --b) {
                    if (acc + hb[b - b0] >= need) break;
                    acc += hb[b - b0];
                }
                s_need = need - acc;
                s_prefix = prefix | ((unsigned)b << shift);
            }
        }
        mask |= 0xffu << shift;
        __syncthreads();
    }
    const unsigned T = s_prefix, need_eq = s_need;  // the k-th best key; how many elements equal to it belong to the result
    // (an odd chunk: thread t walks [t chunk, (t + 1) chunk), and an odd stride between the threads' LDS addresses has no bank conflicts)
    const int chunk = ((n + TOPK_TPB - 1) / TOPK_TPB) | 1, i0 = tid * chunk < n ? tid * chunk : n, i1 = i0 + chunk < n ? i0 + chunk : n;
    unsigned cgt = 0u, ceq = 0u;
    for (int i = i0; i < i1; ++i) {
        const unsigned key = key_at(i);
        cgt += key > T ? 1u : 0u;
        ceq += key == T ? 1u : 0u;
    }
    // exclusive scans of the two counts in index order (thread t owns elements [t chunk, (t + 1) chunk))
    const unsigned sg = wave_incl_scan(cgt, lane), se = wave_incl_scan(ceq, lane);
    if (lane == 63) {
        w_gt[wave] = sg;
        w_eq[wave] = se;
    }
    __syncthreads();
    unsigned pg = sg - cgt, pe = se - ceq;
    for (int w = 0; w < wave; ++w) {
        pg += w_gt[w];
        pe += w_eq[w];
    }
    const unsigned n_gt = (unsigned)k - need_eq;
    for (int i = i0; i < i1; ++i) {
        const unsigned key = key_at(i);
        int slot = -1;
        if (key > T) slot = (int)pg++;
        else if (key == T) {
            if (pe < need_eq) slot = (int)(n_gt + pe);
            ++pe;
        }
        if (slot >= 0) {
            sel_key[slot] = key;
            sel_idx[slot] = i;
            sel_val[slot] = row[i];
        }
    }
    __syncthreads();
    for (int e = tid; e < k; e += TOPK_TPB) {
        const unsigned key = sel_key[e];
        const int idx = sel_idx[e];
        int rank = 0;
        for (int j = 0; j < k; ++j) rank += (sel_key[j] > key || (sel_key[j] == key && sel_idx[j] < idx)) ? 1 : 0;
        values[(int64_t)blockIdx.x * k + rank] = sel_val[e];
        indices[(int64_t)blockIdx.x * k + rank] = (float)idx;  // indices are returned as f32
    }
}

__global__ void range_kernel(float start, float delta, int64_t n, float* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = start + (float)i * delta;  // math.rs:2049-2053
}
__global__ void range_i64_kernel(int64_t start, int64_t delta, int64_t n, int64_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = start + i * delta;
}
template <typename S, typename D>
__global__ void cast_kernel(const S* __restrict__ in, D* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (D)in[i];
}

// f32 -> u8 with Rust's `as u8` (src/tensor.rs:92-97, TensorView::reinterpret_as_u8): truncation toward zero, saturating, NaN -> 0
__global__ void cast_f32_u8_kernel(const float* __restrict__ in, uint8_t* __restrict__ out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = in[i];
        out[i] = (v != v) ? (uint8_t)0 : (uint8_t)fminf(fmaxf(truncf(v), 0.0f), 255.0f);
    }
}

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192)); }

struct Word16 {
    uint32_t a, b, c, d;
};

// Canonicalise the descriptor before launching: drop unit dims, merge dims that are contiguous on both sides, move
// 16-byte words when the innermost dim allows it, and index in 32 bits when everything fits.
int launch_copy(LeleCtx* ctx, const void* in, void* out, int64_t numel, const CopyDesc& d0, size_t esize) {
    if (numel == 0) return 0;
    CopyDesc d{};
    d.ioff = d0.ioff;
    d.ooff = d0.ooff;
    int r = 0;
    for (int k = 0; k < d0.rank; ++k) {
        if (d0.oshape[k] == 1) continue;
        d.oshape[r] = d0.oshape[k];
        d.istride[r] = d0.istride[k];
        d.ostride[r] = d0.ostride[k];
        d.imod[r] = d0.imod[k];
        ++r;
    }
    for (int k = r - 2; k >= 0; --k) {  // merge k with k+1
        if (d.imod[k] == 0 && d.imod[k + 1] == 0 && d.istride[k] == d.istride[k + 1] * d.oshape[k + 1] &&
            d.ostride[k] == d.ostride[k + 1] * d.oshape[k + 1]) {
            d.oshape[k] *= d.oshape[k + 1];
            d.istride[k] = d.istride[k + 1];
            d.ostride[k] = d.ostride[k + 1];
            for (int j = k + 1; j + 1 < r; ++j) {
                d.oshape[j] = d.oshape[j + 1];
                d.istride[j] = d.istride[j + 1];
                d.ostride[j] = d.ostride[j + 1];
                d.imod[j] = d.imod[j + 1];
            }
            --r;
        }
    }
    d.rank = r;
    // 16-byte words
    const int64_t v = (int64_t)(16 / esize);
    bool vec = r >= 1 && d.istride[r - 1] == 1 && d.ostride[r - 1] == 1 && d.imod[r - 1] == 0 && d.oshape[r - 1] % v == 0 &&
               d.ioff % v == 0 && d.ooff % v == 0 && ((uintptr_t)in % 16) == 0 && ((uintptr_t)out % 16) == 0;
    for (int k = 0; vec && k + 1 < r; ++k)
        vec = d.istride[k] % v == 0 && d.ostride[k] % v == 0;
    if (vec) {
        d.oshape[r - 1] /= v;
        d.ioff /= v;
        d.ooff /= v;
        for (int k = 0; k + 1 < r; ++k) {
            d.istride[k] /= v;
            d.ostride[k] /= v;
        }
        numel /= v;
    }
    // 32-bit indexing when every reachable offset fits
    int64_t imax = d.ioff < 0 ? -d.ioff : d.ioff, omax = d.ooff < 0 ? -d.ooff : d.ooff;
    for (int k = 0; k < r; ++k) {
        imax += (d.oshape[k] - 1) * (d.istride[k] < 0 ? -d.istride[k] : d.istride[k]);
        omax += (d.oshape[k] - 1) * (d.ostride[k] < 0 ? -d.ostride[k] : d.ostride[k]);
    }
    const bool i32 = numel < (int64_t(1) << 31) && imax < (int64_t(1) << 31) && omax < (int64_t(1) << 31);
    // transposing copy: inner dim strided in the input, another dim contiguous there -> LDS-tiled kernel
    if (!vec && r >= 2 && d.ostride[r - 1] == 1 && d.istride[r - 1] != 1 && d.oshape[r - 1] >= 8) {
        int tj = -1;
        bool plain = true;
        for (int k = 0; k < r; ++k) plain = plain && d.imod[k] == 0;
        for (int k = 0; plain && k + 1 < r; ++k)
            if (d.istride[k] == 1 && d.oshape[k] >= 8) tj = k;
        int64_t others = 1;
        for (int k = 0; k + 1 < r; ++k)
            if (k != tj) others *= d.oshape[k];
        if (tj >= 0 && others <= 65535) {
            const dim3 tgrid((unsigned)((d.oshape[r - 1] + 63) / 64), (unsigned)((d.oshape[tj] + 63) / 64), (unsigned)others);
            if (tgrid.y <= 65535) {
                if (esize == 8)
                    hipLaunchKernelGGL(transpose_tile_kernel<uint64_t>, tgrid, dim3(256), 0, ctx->stream, (const uint64_t*)in,
                                       (uint64_t*)out, d, tj);
                else
                    hipLaunchKernelGGL(transpose_tile_kernel<uint32_t>, tgrid, dim3(256), 0, ctx->stream, (const uint32_t*)in,
                                       (uint32_t*)out, d, tj);
                LELE_HIP_CHECK(hipGetLastError());
                return 0;
            }
        }
    }
    const dim3 grid(grid_for(numel)), block(256);
#define LELE_COPY(W, I) \
    hipLaunchKernelGGL((strided_copy_kernel<W, I>), grid, block, 0, ctx->stream, (const W*)in, (W*)out, numel, d)
    if (vec) {
        if (i32) LELE_COPY(Word16, int32_t); else LELE_COPY(Word16, int64_t);
    } else if (esize == 8) {
        if (i32) LELE_COPY(uint64_t, int32_t); else LELE_COPY(uint64_t, int64_t);
    } else {
        if (i32) LELE_COPY(uint32_t, int32_t); else LELE_COPY(uint32_t, int64_t);
    }
#undef LELE_COPY
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

std::vector<int64_t> strides_of(const int64_t* shape, int rank) {
    std::vector<int64_t> s(rank, 1);
    for (int i = rank - 2; i >= 0; --i) s[i] = s[i + 1] * shape[i + 1];
    return s;
}

int check_word(const LeleTensor* t, const char* who, size_t* es) {
    *es = dtype_size(t->dtype);
    LELE_REQUIRE(*es == 4 || *es == 8, "%s: only 4- and 8-byte element types are supported", who);
    LELE_REQUIRE(t->rank <= LELE_MAX_RANK, "%s: rank %d exceeds %d", who, t->rank, LELE_MAX_RANK);
    return 0;
}

// image n: `words` words of W from src + n * sp bytes to dst + n * dp bytes.  grid (chunks, images)
template <typename W>
__global__ __launch_bounds__(256) void copy_pitched_kernel(const char* __restrict__ src, char* __restrict__ dst, size_t words, size_t sp,
                                                           size_t dp) {
    const W* s = reinterpret_cast<const W*>(src + (size_t)blockIdx.y * sp);
    W* d = reinterpret_cast<W*>(dst + (size_t)blockIdx.y * dp);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) d[i] = s[i];
}

}  // namespace

extern "C" {

/* Generic strided gather: out[c0..c(r-1)] = in[offset + sum c_k * stride_k] (coordinate taken modulo mod_k when
 * mod_k > 0).  slice / transpose / expand / tile are this call with host-computed descriptors. */
int lele_hip_strided_copy(LeleCtx* ctx, const LeleTensor* x, const int64_t* out_shape_in, const int64_t* strides,
                          const int64_t* mods, int32_t rank, int64_t offset, LeleBuf* out, int64_t* out_shape,
                          int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out && (rank == 0 || (out_shape_in && strides)), "strided_copy: NULL argument");
    LELE_REQUIRE(rank >= 0 && rank <= LELE_MAX_RANK, "strided_copy: bad rank %d", rank);
    size_t es;
    LELE_TRY(check_word(x, "strided_copy", &es));
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    CopyDesc d{};
    d.rank = rank;
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) {
        d.oshape[i] = out_shape_in[i];
        d.istride[i] = strides[i];
        d.imod[i] = mods ? mods[i] : 0;
        n *= out_shape_in[i];
    }
    int64_t st = 1;
    for (int i = rank - 1; i >= 0; --i) {
        d.ostride[i] = st;
        st *= out_shape_in[i];
    }
    d.ioff = offset;
    d.ooff = 0;
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)n * es));
    LELE_TRY(launch_copy(ctx, dx, out->data, n, d, es));
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(out_shape_in, out_shape_in + rank));
}

/* concat, manipulation.rs:108-207 */
int lele_hip_concat(LeleCtx* ctx, const LeleTensor* const* inputs, size_t ninputs, int64_t axis, LeleBuf* out,
                    int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && inputs && ninputs > 0 && out, "concat: NULL argument or no inputs");
    const int rank = inputs[0]->rank;
    size_t es;
    LELE_TRY(check_word(inputs[0], "concat", &es));
    const int ax = (int)(axis < 0 ? rank + axis : axis);
    LELE_REQUIRE(ax >= 0 && ax < rank, "concat: axis %lld out of range", (long long)axis);
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<int64_t> oshape(inputs[0]->shape, inputs[0]->shape + rank);
    oshape[ax] = 0;
    for (size_t i = 0; i < ninputs; ++i) {
        LELE_REQUIRE(inputs[i]->rank == rank && inputs[i]->dtype == inputs[0]->dtype, "concat: rank/dtype mismatch");
        for (int d = 0; d < rank; ++d)
            LELE_REQUIRE(d == ax || inputs[i]->shape[d] == inputs[0]->shape[d], "concat: shape mismatch on dim %d", d);
        oshape[ax] += inputs[i]->shape[ax];
    }
    int64_t n = 1;
    for (int64_t v : oshape) n *= v;
    std::vector<int64_t> ostr = strides_of(oshape.data(), rank);
    LELE_TRY(ctx->arena_reset());
    LELE_TRY(out->reserve((size_t)n * es));
    int64_t pos = 0;
    for (size_t i = 0; i < ninputs; ++i) {
        const LeleTensor* t = inputs[i];
        const void* dx = nullptr;
        LELE_TRY(ctx->dev_ptr(t, &dx));
        CopyDesc d{};
        d.rank = rank;
        std::vector<int64_t> istr = strides_of(t->shape, rank);
        for (int k = 0; k < rank; ++k) {
            d.oshape[k] = t->shape[k];
            d.istride[k] = istr[k];
            d.ostride[k] = ostr[k];
        }
        d.ooff = pos * ostr[ax];
        LELE_TRY(launch_copy(ctx, dx, out->data, numel(t), d, es));
        pos += t->shape[ax];
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

/* pad, manipulation.rs:382-587: pads = [begin_0..begin_r-1, end_0..end_r-1] (already expanded to 2*rank by the host
 * mirror, negatives clamped to 0); mode 0 constant / 1 edge / 2 reflect; fill = raw 4- or 8-byte pattern */
int lele_hip_pad(LeleCtx* ctx, const LeleTensor* x, const int64_t* pads, int32_t mode, uint64_t fill_bits, LeleBuf* out,
                 int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && pads && out, "pad: NULL argument");
    LELE_REQUIRE(mode >= 0 && mode <= 2, "pad: unknown mode %d", mode);
    size_t es;
    LELE_TRY(check_word(x, "pad", &es));
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int rank = x->rank;
    PadDesc d{};
    d.rank = rank;
    d.mode = mode;
    std::vector<int64_t> oshape(rank);
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) {
        const int64_t b = pads[i] < 0 ? 0 : pads[i], e = pads[rank + i] < 0 ? 0 : pads[rank + i];
        d.before[i] = b;
        d.ishape[i] = x->shape[i];
        d.oshape[i] = oshape[i] = x->shape[i] + b + e;
        if (mode == 2) LELE_REQUIRE(b < x->shape[i] && e <= x->shape[i], "pad: reflect padding must be smaller than the dim");
        n *= oshape[i];
    }
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)n * es));
    if (n) {
        if (es == 8)
            hipLaunchKernelGGL(pad_kernel<uint64_t>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const uint64_t*)dx,
                               (uint64_t*)out->data, n, d, (uint64_t)fill_bits);
        else
            hipLaunchKernelGGL(pad_kernel<uint32_t>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const uint32_t*)dx,
                               (uint32_t*)out->data, n, d, (uint32_t)fill_bits);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

/* gather, manipulation.rs:589-641: indices may be f32 (values as floats), i64 or i32 */
int lele_hip_gather(LeleCtx* ctx, const LeleTensor* data, const LeleTensor* indices, int64_t axis, LeleBuf* out,
                    int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && data && indices && out, "gather: NULL argument");
    size_t es;
    LELE_TRY(check_word(data, "gather", &es));
    const int rank = data->rank;
    const int ax = (int)(axis < 0 ? rank + axis : axis);
    LELE_REQUIRE(ax >= 0 && ax < rank, "gather: axis %lld out of range", (long long)axis);
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<int64_t> oshape(data->shape, data->shape + ax);
    oshape.insert(oshape.end(), indices->shape, indices->shape + indices->rank);
    oshape.insert(oshape.end(), data->shape + ax + 1, data->shape + rank);
    LELE_REQUIRE(oshape.size() <= LELE_MAX_RANK, "gather: result rank exceeds %d", LELE_MAX_RANK);
    int64_t outer = 1, inner = 1;
    for (int i = 0; i < ax; ++i) outer *= data->shape[i];
    for (int i = ax + 1; i < rank; ++i) inner *= data->shape[i];
    const int64_t nidx = numel(indices), total = outer * nidx * inner;
    if (indices->mem != LELE_MEM_DEVICE && data->shape[ax] > 0) {  // host-visible indices: the reference's panic, before any launch
        for (int64_t j = 0; j < nidx; ++j) {
            int64_t v = indices->dtype == LELE_F32 ? (int64_t)((const float*)indices->data)[j]
                        : indices->dtype == LELE_I64 ? ((const int64_t*)indices->data)[j] : (int64_t)((const int32_t*)indices->data)[j];
            if (v < 0) v += data->shape[ax];
            LELE_REQUIRE(v >= 0 && v < data->shape[ax], "gather: index %lld (element %lld) is out of range for axis %d of size %lld",
                         (long long)v, (long long)j, ax, (long long)data->shape[ax]);
        }
    }
    LELE_TRY(ctx->arena_reset());
    const void *dd = nullptr, *di = nullptr;
    LELE_TRY(ctx->dev_ptr(data, &dd));
    LELE_TRY(ctx->dev_ptr(indices, &di));
    LELE_TRY(out->reserve((size_t)total * es));
    if (total) {
        const dim3 g(grid_for(total)), b(256);
#define LELE_GATHER(W, I)                                                                                          \
    hipLaunchKernelGGL((gather_kernel<W, I>), g, b, 0, ctx->stream, (const W*)dd, (const I*)di, (W*)out->data, outer, \
                       data->shape[ax], inner, nidx, ctx->deverr_dev)
        if (es == 8) {
            if (indices->dtype == LELE_F32) LELE_GATHER(uint64_t, float);
            else if (indices->dtype == LELE_I64) LELE_GATHER(uint64_t, int64_t);
            else LELE_GATHER(uint64_t, int32_t);
        } else {
            if (indices->dtype == LELE_F32) LELE_GATHER(uint32_t, float);
            else if (indices->dtype == LELE_I64) LELE_GATHER(uint32_t, int64_t);
            else LELE_GATHER(uint32_t, int32_t);
        }
#undef LELE_GATHER
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

int lele_hip_gather_elements(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* indices, int64_t axis, LeleBuf* out,
                             int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && indices && out, "gather_elements: NULL argument");
    LELE_REQUIRE(x->dtype == LELE_F32 && indices->dtype == LELE_F32, "gather_elements: f32 data and f32-carried indices");
    LELE_REQUIRE(x->rank == indices->rank && x->rank <= LELE_MAX_RANK, "gather_elements: rank mismatch");
    const int rank = x->rank;
    const int ax = (int)(axis < 0 ? rank + axis : axis);
    LELE_REQUIRE(ax >= 0 && ax < rank, "gather_elements: axis out of range");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    GeDesc d{};
    d.rank = rank;
    d.axis = ax;
    d.axis_dim = x->shape[ax];
    std::vector<int64_t> xs = strides_of(x->shape, rank);
    for (int i = 0; i < rank; ++i) {
        d.ishape[i] = indices->shape[i];
        d.xstride[i] = xs[i];
    }
    const int64_t total = numel(indices);
    if (indices->mem != LELE_MEM_DEVICE && d.axis_dim > 0) {
        for (int64_t j = 0; j < total; ++j) {
            int64_t v = (int64_t)((const float*)indices->data)[j];
            if (v < 0) v += d.axis_dim;
            LELE_REQUIRE(v >= 0 && v < d.axis_dim, "gather_elements: index %lld (element %lld) is out of range for axis %d of size %lld",
                         (long long)v, (long long)j, ax, (long long)d.axis_dim);
        }
    }
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *di = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(indices, &di));
    LELE_TRY(out->reserve((size_t)total * 4));
    if (total) {
        hipLaunchKernelGGL(gather_elements_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float*)dx,
                           (const float*)di, (float*)out->data, total, d, ctx->deverr_dev);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(indices->shape, indices->shape + rank));
}

/* adaptive_avg_pool1d, pooling.rs:1-30.  Upstream takes flat slices + (channels, input_len, output_len); here x is
 * [.., L] (all leading dims are channels) and the result [.., output_len]. */
int lele_hip_adaptive_avg_pool1d(LeleCtx* ctx, const LeleTensor* x, int64_t output_len, LeleBuf* out, int64_t* out_shape,
                                 int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "adaptive_avg_pool1d: NULL argument");
    LELE_REQUIRE(x->dtype == LELE_F32 && x->rank >= 1, "adaptive_avg_pool1d: f32 input of rank >= 1 required");
    LELE_REQUIRE(output_len >= 0, "adaptive_avg_pool1d: negative output length");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t in_len = x->shape[x->rank - 1];
    int64_t channels = 1;
    for (int i = 0; i + 1 < x->rank; ++i) channels *= x->shape[i];
    std::vector<int64_t> oshape(x->shape, x->shape + x->rank);
    oshape.back() = output_len;
    const int64_t total = channels * output_len;
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)total * 4));
    if (total) {
        hipLaunchKernelGGL(adaptive_avg_pool1d_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float*)dx,
                           (float*)out->data, channels, in_len, output_len);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

// x_pitch of a one-operand op: 0 = dense; otherwise the operand must be a device tensor and the pitch must cover one image
static int check_x_pitch(const LeleTensor* x, const LelePitch* pv, int64_t per_image, const char* what) {
    if (!pv) return 0;
    LELE_REQUIRE(pv->y_pitch == 0, "%s: y_pitch must be 0 (one tensor operand)", what);
    LELE_REQUIRE(pv->x_pitch == 0 || (x->mem == LELE_MEM_DEVICE && pv->x_pitch >= per_image),
                 "%s: x_pitch needs a device tensor and must cover one image", what);
    return 0;
}

/* resize_nearest, conv2d.rs:1261-1382: output H, W already resolved from sizes / scales by the host mirror */
static int resize_nearest_entry(LeleCtx* ctx, const LeleTensor* x, int64_t out_h, int64_t out_w, int asymmetric, const LelePitch* pv,
                                LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "resize_nearest: NULL argument");
    LELE_REQUIRE(x->rank == 4 && x->dtype == LELE_F32, "Resize: expected rank-4 input");
    LELE_REQUIRE(out_h > 0 && out_w > 0, "Resize: output dimensions must be positive, got out_h=%lld out_w=%lld",
                 (long long)out_h, (long long)out_w);
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t planes = x->shape[0] * x->shape[1], total = planes * out_h * out_w;
    const int64_t in_img = x->shape[1] * x->shape[2] * x->shape[3], out_img = x->shape[1] * out_h * out_w;
    LELE_TRY(check_x_pitch(x, pv, in_img, "resize_nearest_pitched"));
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    float* dst = nullptr;
    if (pv) {
        LELE_TRY(lele::pitched_out(out, pv, x->shape[0], out_img, 4, (void**)&dst));
    } else {
        LELE_TRY(out->reserve((size_t)total * 4));
        dst = (float*)out->data;
    }
    const long long xbs_ = pv && pv->x_pitch ? pv->x_pitch : in_img, obs_ = pv && pv->out_pitch ? pv->out_pitch : out_img;
    const int64_t up = x->shape[2] > 0 && out_h % x->shape[2] == 0 ? out_h / x->shape[2] : 0;
    if (total && asymmetric && (up == 2 || up == 4 || up == 8) && out_w == up * x->shape[3] && x->shape[3] % 4 == 0 && x->shape[0] <= 65535 &&
        in_img / 4 < (int64_t(1) << 32) && ((((uintptr_t)dx) | ((uintptr_t)dst)) & 15) == 0 && xbs_ % 4 == 0 && obs_ % 4 == 0) {
        const unsigned quads = (unsigned)(in_img / 4);
        const dim3 ugrid((quads + 255u) / 256u, (unsigned)x->shape[0]);
        if (up == 2)
            hipLaunchKernelGGL(resize_up_kernel<2>, ugrid, dim3(256), 0, ctx->stream, (const float*)dx, dst, quads, (int)(x->shape[3] / 4), xbs_, obs_);
        else if (up == 4)
            hipLaunchKernelGGL(resize_up_kernel<4>, ugrid, dim3(256), 0, ctx->stream, (const float*)dx, dst, quads, (int)(x->shape[3] / 4), xbs_, obs_);
        else
            hipLaunchKernelGGL(resize_up_kernel<8>, ugrid, dim3(256), 0, ctx->stream, (const float*)dx, dst, quads, (int)(x->shape[3] / 4), xbs_, obs_);
        LELE_HIP_CHECK(hipGetLastError());
    } else if (total) {
        hipLaunchKernelGGL(resize_nearest_kernel, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float*)dx, dst, planes,
                           (int)x->shape[2], (int)x->shape[3], (int)out_h, (int)out_w, asymmetric, (int)x->shape[1],
                           (long long)(pv && pv->x_pitch ? pv->x_pitch : in_img), (long long)(pv && pv->out_pitch ? pv->out_pitch : out_img));
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape(out_shape, out_rank, {x->shape[0], x->shape[1], out_h, out_w});
}
int lele_hip_resize_nearest(LeleCtx* ctx, const LeleTensor* x, int64_t out_h, int64_t out_w, int asymmetric,
                            LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    return resize_nearest_entry(ctx, x, out_h, out_w, asymmetric, nullptr, out, out_shape, out_rank);
}
int lele_hip_resize_nearest_pitched(LeleCtx* ctx, const LeleTensor* x, int64_t out_h, int64_t out_w, int asymmetric,
                                    const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(pitch, "resize_nearest_pitched: pitch is NULL");
    return resize_nearest_entry(ctx, x, out_h, out_w, asymmetric, pitch, out, out_shape, out_rank);
}

/* max_pool2d, conv2d.rs:1051-1254 */
static int max_pool2d_entry(LeleCtx* ctx, const LeleTensor* x, const int64_t* kernel_shape, size_t nk, const int64_t* strides, size_t ns,
                            const int64_t* pads, size_t np, const int64_t* dilations, size_t nd, int ceil_mode, const LelePitch* pv,
                            LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out && kernel_shape && nk >= 1, "max_pool2d: NULL argument");
    LELE_REQUIRE(x->rank == 4 && x->dtype == LELE_F32, "MaxPool2d: expected rank-4 input");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    PoolDesc d{};
    d.in_h = (int)x->shape[2];
    d.in_w = (int)x->shape[3];
    d.kh = (int)kernel_shape[0];
    d.kw = nk > 1 ? (int)kernel_shape[1] : d.kh;
    d.sh = ns == 0 ? 1 : (int)strides[0];
    d.sw = ns > 1 ? (int)strides[1] : d.sh;
    d.pt = np == 0 ? 0 : (int)pads[0];
    d.pl = np > 1 ? (int)pads[1] : d.pt;
    const int pb = np > 2 ? (int)pads[2] : d.pt, pr = np > 3 ? (int)pads[3] : d.pl;
    d.dh = nd == 0 ? 1 : (int)dilations[0];
    d.dw = nd > 1 ? (int)dilations[1] : d.dh;
    const int ekh = d.dh * (d.kh - 1) + 1, ekw = d.dw * (d.kw - 1) + 1;
    const int64_t num_h = d.in_h + d.pt + pb - ekh, num_w = d.in_w + d.pl + pr - ekw;
    LELE_REQUIRE(num_h >= 0 && num_w >= 0 && d.sh > 0 && d.sw > 0, "max_pool2d: window larger than the padded input");
    d.out_h = (int)(ceil_mode ? (num_h + d.sh - 1) / d.sh + 1 : num_h / d.sh + 1);
    d.out_w = (int)(ceil_mode ? (num_w + d.sw - 1) / d.sw + 1 : num_w / d.sw + 1);
    const int64_t planes = x->shape[0] * x->shape[1], total = planes * d.out_h * d.out_w;
    const int64_t plane_in = (int64_t)d.in_h * d.in_w, plane_out = (int64_t)d.out_h * d.out_w;
    const int64_t in_img = x->shape[1] * plane_in, out_img = x->shape[1] * plane_out;
    LELE_TRY(check_x_pitch(x, pv, in_img, "max_pool2d_pitched"));
    d.channels = (int)x->shape[1];
    d.xbs = pv && pv->x_pitch ? pv->x_pitch : in_img;
    d.obs = pv && pv->out_pitch ? pv->out_pitch : out_img;
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    float* dst = nullptr;
    if (pv) {
        LELE_TRY(lele::pitched_out(out, pv, x->shape[0], out_img, 4, (void**)&dst));
    } else {
        LELE_TRY(out->reserve((size_t)total * 4));
        dst = (float*)out->data;
    }
    if (total) {
        const int64_t in_total = planes * plane_in;
        // whole planes in LDS when (input + output plane) fit 48 KB: as many planes per workgroup as keep >= 4 workgroups per CU
        // windows of more than kh + kw + 2 cells go row-wise, then column-wise (kh + kw reads per output): + in_h x out_w floats a plane
        const bool separable = (int64_t)d.kh * d.kw > (int64_t)d.kh + d.kw + 2;
        const int64_t per_plane = ((plane_in + 3) & ~int64_t(3)) + ((plane_out + 3) & ~int64_t(3)) + (separable ? (int64_t)d.in_h * d.out_w : 0);
        int64_t ppb = (12 * 1024) / std::max<int64_t>(per_plane, 1);
        ppb = std::min<int64_t>(ppb, d.channels);
        while (ppb > 1 && x->shape[0] * ((d.channels + ppb - 1) / ppb) < 4 * (int64_t)ctx->num_cus) ppb = (ppb + 1) / 2;
        if (ppb >= 1 && x->shape[0] <= 65535 && in_img < (int64_t(1) << 31) && out_img < (int64_t(1) << 31) && plane_in % 4 == 0 &&
            plane_out % 4 == 0) {
            // plane sizes that are multiples of four floats keep every plane 16-byte aligned inside the LDS runs
            const size_t lds = (size_t)(((ppb * plane_in + 3) & ~int64_t(3)) + ppb * plane_out + (separable ? ppb * d.in_h * d.out_w : 0)) * 4;
            hipLaunchKernelGGL(max_pool2d_lds_kernel, dim3((unsigned)((d.channels + ppb - 1) / ppb), (unsigned)x->shape[0]), dim3(256), lds,
                               ctx->stream, (const float*)dx, dst, d, (int)ppb, (int)separable);
        } else if (total < (int64_t(1) << 31) && in_total < (int64_t(1) << 31)) {
            hipLaunchKernelGGL(max_pool2d_kernel<int32_t>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float*)dx, dst, planes, d);
        } else {
            hipLaunchKernelGGL(max_pool2d_kernel<int64_t>, dim3(grid_for(total)), dim3(256), 0, ctx->stream, (const float*)dx, dst, planes, d);
        }
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape(out_shape, out_rank, {x->shape[0], x->shape[1], (int64_t)d.out_h, (int64_t)d.out_w});
}
int lele_hip_max_pool2d(LeleCtx* ctx, const LeleTensor* x, const int64_t* kernel_shape, size_t nk,
                        const int64_t* strides, size_t ns, const int64_t* pads, size_t np, const int64_t* dilations,
                        size_t nd, int ceil_mode, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    return max_pool2d_entry(ctx, x, kernel_shape, nk, strides, ns, pads, np, dilations, nd, ceil_mode, nullptr, out, out_shape, out_rank);
}
int lele_hip_max_pool2d_pitched(LeleCtx* ctx, const LeleTensor* x, const int64_t* kernel_shape, size_t nk, const int64_t* strides,
                                size_t ns, const int64_t* pads, size_t np, const int64_t* dilations, size_t nd, int ceil_mode,
                                const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(pitch, "max_pool2d_pitched: pitch is NULL");
    return max_pool2d_entry(ctx, x, kernel_shape, nk, strides, ns, pads, np, dilations, nd, ceil_mode, pitch, out, out_shape, out_rank);
}

/* [N, C, P...] -> [N, P, C] of a channel view into a row window: the detection tail's Concat(levels) -> Transpose -> Split(heads)
 * as one transposing copy per (level, head) -- see lele_hip.h */
int lele_hip_transpose_cp_pitched(LeleCtx* ctx, const LeleTensor* x, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out && pitch, "transpose_cp_pitched: NULL argument");
    LELE_REQUIRE(x->rank >= 3 && x->dtype == LELE_F32, "transpose_cp_pitched: an f32 tensor [N, C, ...] of rank >= 3 required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t images = x->shape[0], c = x->shape[1];
    int64_t pos = 1;
    for (int d = 2; d < x->rank; ++d) pos *= x->shape[d];
    const int64_t per = c * pos;
    LELE_TRY(check_x_pitch(x, pitch, per, "transpose_cp_pitched"));
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    void* dst = nullptr;
    LELE_TRY(lele::pitched_out(out, pitch, images, per, 4, &dst));
    if (images * per) {
        LELE_REQUIRE(images <= 65535 && (c + 31) / 32 <= 65535 && pos < (int64_t(1) << 31) && c < (int64_t(1) << 31), "transpose_cp_pitched: tensor too large");
        hipLaunchKernelGGL(transpose_cp_kernel, dim3((unsigned)((pos + 31) / 32), (unsigned)((c + 31) / 32), (unsigned)images), dim3(256), 0, ctx->stream,
                           (const float*)dx, (float*)dst, (int)c, (int)pos, (long long)(pitch->x_pitch ? pitch->x_pitch : per),
                           (long long)(pitch->out_pitch ? pitch->out_pitch : per));
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape(out_shape, out_rank, {images, pos, c});
}

/* Copy between channel views (or a view and a dense tensor): image n of `x` (x_pitch apart, or dense) -> image n of the destination
 * window (out_offset / out_pitch, or a dense, resized `out`).  What Concat / Split along C are when neither side can work in place;
 * any element type.  The result has x's shape. */
int lele_hip_copy_pitched(LeleCtx* ctx, const LeleTensor* x, const LelePitch* pitch, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out && pitch, "copy_pitched: NULL argument");
    LELE_REQUIRE(x->rank >= 1, "copy_pitched: rank >= 1 required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t es = dtype_size(x->dtype);
    const int64_t images = x->shape[0], per = images ? numel(x) / images : 0;
    LELE_TRY(check_x_pitch(x, pitch, per, "copy_pitched"));
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    void* dst = nullptr;
    LELE_TRY(lele::pitched_out(out, pitch, images, per, es, &dst));
    if (images * per) {
        LELE_REQUIRE(images <= 65535, "copy_pitched: more than 65535 images");
        const size_t row = (size_t)per * es, sp = (size_t)(pitch->x_pitch ? pitch->x_pitch : per) * es,
                     dp = (size_t)(pitch->out_pitch ? pitch->out_pitch : per) * es;
        const int w = ((((uintptr_t)dx) | ((uintptr_t)dst) | row | sp | dp) & 15) == 0 ? 16
                      : ((((uintptr_t)dx) | ((uintptr_t)dst) | row | sp | dp) & 3) == 0 ? 4 : 1;
        const unsigned chunks = (unsigned)std::max<size_t>(1, std::min<size_t>((row / w + 2047) / 2048, 4096));
        const dim3 cgrid(chunks, (unsigned)images);
        if (w == 16)
            hipLaunchKernelGGL(copy_pitched_kernel<uint4>, cgrid, dim3(256), 0, ctx->stream, (const char*)dx, (char*)dst, row / 16, sp, dp);
        else if (w == 4)
            hipLaunchKernelGGL(copy_pitched_kernel<unsigned>, cgrid, dim3(256), 0, ctx->stream, (const char*)dx, (char*)dst, row / 4, sp, dp);
        else
            hipLaunchKernelGGL(copy_pitched_kernel<unsigned char>, cgrid, dim3(256), 0, ctx->stream, (const char*)dx, (char*)dst, row, sp, dp);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

/* topk, conv2d.rs:1385-1435: last axis only (the `axis` argument is ignored by the reference too) */
int lele_hip_topk(LeleCtx* ctx, const LeleTensor* x, int64_t k, int largest, LeleBuf* out_values, LeleBuf* out_indices,
                  int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out_values && out_indices, "topk: NULL argument");
    LELE_REQUIRE(x->rank >= 1 && x->dtype == LELE_F32, "topk: f32 tensor of rank >= 1 required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = x->shape[x->rank - 1];
    const int64_t rows = n ? numel(x) / n : 0;
    const int64_t kk = std::min<int64_t>(std::max<int64_t>(k, 0), n);  // k.min(last_dim)
    std::vector<int64_t> oshape(x->shape, x->shape + x->rank);
    oshape.back() = kk;
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out_values->reserve((size_t)rows * kk * 4));
    LELE_TRY(out_indices->reserve((size_t)rows * kk * 4));
    if (rows * kk) {
        const dim3 tgrid((unsigned)((n + 255) / 256), (unsigned)rows);
        if (n > 1024 && kk <= 1024 && n < (int64_t(1) << 31) && rows < (int64_t(1) << 31)) {  // long rows: radix select, a workgroup per row
            if (n <= TOPK_STAGE_MAX) {
                auto kern = topk_select_kernel<true>;
                LELE_HIP_CHECK(lele::ensure_dyn_lds(reinterpret_cast<const void*>(kern), TOPK_STAGE_MAX * 4));  // (the opt-in is remembered per kernel: ask for the most)
                hipLaunchKernelGGL(kern, dim3((unsigned)rows), dim3(TOPK_TPB), (size_t)n * 4, ctx->stream, (const float*)dx, n, (int)kk, largest,
                                   (float*)out_values->data, (float*)out_indices->data);
            } else {
                hipLaunchKernelGGL(topk_select_kernel<false>, dim3((unsigned)rows), dim3(TOPK_TPB), 0, ctx->stream, (const float*)dx, n, (int)kk,
                                   largest, (float*)out_values->data, (float*)out_indices->data);
            }
        } else if (n <= 4096 || kk > 1024 || n >= (int64_t(1) << 31)) {
            hipLaunchKernelGGL(topk_kernel, tgrid, dim3(256), 0, ctx->stream, (const float*)dx, n, kk, largest,
                               (float*)out_values->data, (float*)out_indices->data);
        } else {  // prefilter on the first 2048 elements, then rank the survivors only
            void *t0 = nullptr, *cnt = nullptr, *cv = nullptr, *ci = nullptr;
            LELE_TRY(ctx->arena_alloc((size_t)rows * 4, &t0));
            LELE_TRY(ctx->arena_alloc((size_t)rows * 4, &cnt));
            LELE_TRY(ctx->arena_alloc((size_t)rows * n * 4, &cv));
            LELE_TRY(ctx->arena_alloc((size_t)rows * n * 4, &ci));
            hipLaunchKernelGGL(topk_init_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, ctx->stream, rows, largest,
                               (float*)t0, (int*)cnt);
            hipLaunchKernelGGL(topk_threshold_kernel, dim3(32, (unsigned)rows), dim3(256), 0, ctx->stream, (const float*)dx, n, kk,
                               largest, (float*)t0, (int*)cnt);
            hipLaunchKernelGGL(topk_compact_kernel, tgrid, dim3(256), 0, ctx->stream, (const float*)dx, n, largest,
                               (const float*)t0, (int*)cnt, (float*)cv, (int*)ci);
            const dim3 rgrid((unsigned)((n + 63) / 64), (unsigned)rows);
            hipLaunchKernelGGL(topk_rank_kernel, rgrid, dim3(256), 0, ctx->stream, (const float*)cv, (const int*)ci, n, kk,
                               largest, (const int*)cnt, (float*)out_values->data, (float*)out_indices->data);
        }
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

/* range, math.rs:2033-2082: n and the first/step values are resolved by the host mirror (they are scalars) */
int lele_hip_range_f32(LeleCtx* ctx, float start, float delta, int64_t n, LeleBuf* out, int64_t* out_shape,
                       int32_t* out_rank) {
    LELE_REQUIRE(ctx && out && n >= 0, "range: bad argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(out->reserve((size_t)n * 4));
    if (n) {
        hipLaunchKernelGGL(range_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, start, delta, n, (float*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape(out_shape, out_rank, {n});
}
int lele_hip_range_i64(LeleCtx* ctx, int64_t start, int64_t delta, int64_t n, LeleBuf* out, int64_t* out_shape,
                       int32_t* out_rank) {
    LELE_REQUIRE(ctx && out && n >= 0, "range_i64: bad argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(out->reserve((size_t)n * 8));
    if (n) {
        hipLaunchKernelGGL(range_i64_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, start, delta, n,
                           (int64_t*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape(out_shape, out_rank, {n});
}

/* constant_of_shape, shape.rs:122-135: fill with a raw 4- or 8-byte pattern */
int lele_hip_fill(LeleCtx* ctx, const int64_t* shape, int32_t rank, int32_t dtype, uint64_t bits, LeleBuf* out,
                  int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && out && (rank == 0 || shape) && rank >= 0 && rank <= LELE_MAX_RANK, "fill: bad argument");
    const size_t es = dtype_size(dtype);
    LELE_REQUIRE(es == 4 || es == 8, "fill: 4- or 8-byte element types only");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t n = 1;
    for (int i = 0; i < rank; ++i) n *= shape[i];
    LELE_TRY(out->reserve((size_t)n * es));
    if (n) {
        if (es == 8)
            hipLaunchKernelGGL(fill_kernel<uint64_t>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (uint64_t*)out->data,
                               n, (uint64_t)bits);
        else
            hipLaunchKernelGGL(fill_kernel<uint32_t>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (uint32_t*)out->data,
                               n, (uint32_t)bits);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(shape, shape + rank));
}

/* cast_to_f32 / cast_to_i64, utils.rs:66-101 */
int lele_hip_cast(LeleCtx* ctx, const LeleTensor* x, int32_t to_dtype, LeleBuf* out, int64_t* out_shape,
                  int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "cast: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = numel(x);
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)n * dtype_size(to_dtype)));
    const dim3 g(grid_for(n)), b(256);
    if (n) {
#define LELE_CAST(S, D) hipLaunchKernelGGL((cast_kernel<S, D>), g, b, 0, ctx->stream, (const S*)dx, (D*)out->data, n)
        if (x->dtype == LELE_F32 && to_dtype == LELE_I64) LELE_CAST(float, int64_t);
        else if (x->dtype == LELE_I64 && to_dtype == LELE_F32) LELE_CAST(int64_t, float);
        else if (x->dtype == LELE_I32 && to_dtype == LELE_F32) LELE_CAST(int32_t, float);
        else if (x->dtype == LELE_I32 && to_dtype == LELE_I64) LELE_CAST(int32_t, int64_t);
        else if (x->dtype == LELE_F32 && to_dtype == LELE_F32) LELE_CAST(float, float);
        else if (x->dtype == LELE_I64 && to_dtype == LELE_I64) LELE_CAST(int64_t, int64_t);
        else if (x->dtype == LELE_U8 && to_dtype == LELE_F32) LELE_CAST(uint8_t, float);
        else if (x->dtype == LELE_I8 && to_dtype == LELE_F32) LELE_CAST(int8_t, float);
        else if (x->dtype == LELE_U8 && to_dtype == LELE_I64) LELE_CAST(uint8_t, int64_t);
        else if (x->dtype == LELE_I8 && to_dtype == LELE_I64) LELE_CAST(int8_t, int64_t);
        else if (x->dtype == LELE_F32 && to_dtype == LELE_U8)
            hipLaunchKernelGGL(cast_f32_u8_kernel, g, b, 0, ctx->stream, (const float*)dx, (uint8_t*)out->data, n);
        else {
            set_error("cast: unsupported conversion %d -> %d", x->dtype, to_dtype);
            return 2;
        }
#undef LELE_CAST
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

}  // extern "C"
