// eltwise.hip -- activations, element-wise broadcast ops, reductions and the normalisation kernels.
//
// Unary (src/kernels/math.rs:893-1104, 1280-1479, 2084-2153 -> src/kernels/avx/math.rs:11-603):
//   lele's x86 kernels run an 8-wide AVX2 body (degree-6 polynomial exp with hi/lo ln2 split, A&S erf, ...) over the
//   first len&~7 elements and call libm for the scalar tail.  The device applies THE SAME polynomial, with the same
//   FMA placement, to exactly those elements (bit-exact), and the accurate device libm to the tail elements.
// Binary broadcast / compare / where / clip / prelu (math.rs:69-264, 414-836, 838, 1106, 1163-1278, 1481, 1922-2031):
//   one IEEE operation per element, numpy-style broadcasting -> bit-exact.
// Reductions (math.rs:1527-1920): every output element accumulates its inputs in row-major input order, as the
//   reference's coordinate walk does -> bit-exact.
// LayerNorm / Softmax / RMSNorm / BatchNorm (norm.rs:8-506 -> avx/norm.rs:10-345): the reference's 4x8-lane
//   accumulators, merge order, 8-wide remainder, horizontal sum and scalar tail are reproduced by giving 32 lanes the
//   roles of the 32 SIMD accumulator slots, so the statistics -- and therefore the outputs -- are bit-exact.
#include "common.h"

#include "norm_core.h"
#include "simd_math.h"

#include <math.h>

using namespace lele;

namespace {

enum UnaryOp {
    U_EXP = 0, U_SIGMOID = 1, U_TANH = 2, U_SILU = 3, U_ERF = 4, U_GELU = 5, U_FAST_GELU = 6, U_RELU = 7, U_SQRT = 8,
    U_LOG = 9, U_SIN = 10, U_COS = 11, U_NEG = 12, U_RECIPROCAL = 13, U_SOFTPLUS = 14, U_NOT = 15, U_ABS = 16,
    U_FLOOR = 17, U_CEIL = 18
};

__device__ __forceinline__ float unary_apply(int op, float x, bool body) {
    switch (op) {
        case U_EXP: return body ? exp_poly(x) : expf(x);
        case U_SIGMOID: return body ? sigmoid_poly(x) : 1.0f / (1.0f + expf(-x));
        case U_TANH: return body ? tanh_poly(x) : tanhf(x);
        case U_SILU: return body ? x * sigmoid_poly(x) : x / (1.0f + expf(-x));
        case U_ERF: return body ? erf_poly(x) : erff(x);
        case U_GELU:
            return body ? (x * 0.5f) * (1.0f + erf_poly(x * 0.7071067811865475f))
                        : x * 0.5f * (1.0f + erff(x * 0.7071067811865475f));
        case U_FAST_GELU: {
            if (body) {
                const float x3 = (x * x) * x;
                const float inner = 0.7978845608028654f * fmaf_(0.044715f, x3, x);
                return (x * 0.5f) * (1.0f + tanh_poly(inner));
            }
            const float inner = 0.7978845608028654f * (x + 0.044715f * x * x * x);
            return 0.5f * x * (1.0f + tanhf(inner));
        }
        case U_RELU: return x > 0.0f ? x : 0.0f;
        case U_SQRT: return sqrtf(x);
        case U_LOG: return logf(x);
        case U_SIN: return sinf(x);
        case U_COS: return cosf(x);
        case U_NEG: return -x;
        case U_RECIPROCAL: return 1.0f / x;
        case U_SOFTPLUS: return x > 20.0f ? x : logf(1.0f + expf(x));  // math.rs:1046-1056
        case U_NOT: return x == 0.0f ? 1.0f : 0.0f;                     // math.rs:1508-1525
        case U_ABS: return fabsf(x);
        case U_FLOOR: return floorf(x);
        case U_CEIL: return ceilf(x);
    }
    return x;
}

__global__ void unary_kernel(int op, const float* __restrict__ x, float* __restrict__ y, int64_t len) {
    const int64_t body_end = len & ~int64_t(7);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = unary_apply(op, x[i], i < body_end);
}
// 16-byte variant (both pointers 16-B aligned): a float4 never straddles the 8-aligned body/tail boundary
__global__ void unary_vec4_kernel(int op, const float* __restrict__ x, float* __restrict__ y, int64_t len) {
    const int64_t body_end = len & ~int64_t(7), nvec = len >> 2;
    const float4* xv = reinterpret_cast<const float4*>(x);
    float4* yv = reinterpret_cast<float4*>(y);
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, gstride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = gtid; i < nvec; i += gstride) {
        const float4 v = xv[i];
        const bool body = 4 * i < body_end;
        yv[i] = make_float4(unary_apply(op, v.x, body), unary_apply(op, v.y, body), unary_apply(op, v.z, body),
                            unary_apply(op, v.w, body));
    }
    for (int64_t i = 4 * nvec + gtid; i < len; i += gstride) y[i] = unary_apply(op, x[i], i < body_end);
}

// ------------------------------------------------------------------------------------------ binary broadcast
enum BinaryOp {
    B_ADD = 0, B_SUB = 1, B_MUL = 2, B_DIV = 3, B_POW = 4, B_MAX = 5, B_MIN = 6, B_EQUAL = 7, B_LESS = 8,
    B_GREATER = 9, B_PRELU = 10, B_MOD = 11, B_AND = 12, B_OR = 13
};

struct Bcast {
    int rank;
    int64_t oshape[LELE_MAX_RANK];
    int64_t astride[LELE_MAX_RANK];  // 0 on broadcast dimensions
    int64_t bstride[LELE_MAX_RANK];
    int64_t cstride[LELE_MAX_RANK];  // third operand (where_op)
};

template <typename T>
__device__ __forceinline__ T binary_apply(int op, T a, T b);
template <>
__device__ __forceinline__ float binary_apply<float>(int op, float a, float b) {
    switch (op) {
        case B_ADD: return a + b;
        case B_SUB: return a - b;
        case B_MUL: return a * b;
        case B_DIV: return a / b;
        case B_POW: return powf(a, b);
        case B_MAX: return fmaxf(a, b);
        case B_MIN: return fminf(a, b);
        case B_EQUAL: return a == b ? 1.0f : 0.0f;
        case B_LESS: return a < b ? 1.0f : 0.0f;
        case B_GREATER: return a > b ? 1.0f : 0.0f;
        case B_PRELU: return a < 0.0f ? a * b : a;                        // math.rs:2012-2031
        case B_MOD: return b == 0.0f ? 0.0f : a - b * floorf(a / b);      // math.rs:1163-1192
        case B_AND: return (a != 0.0f && b != 0.0f) ? 1.0f : 0.0f;
        case B_OR: return (a != 0.0f || b != 0.0f) ? 1.0f : 0.0f;
    }
    return a;
}
template <>
__device__ __forceinline__ int64_t binary_apply<int64_t>(int op, int64_t a, int64_t b) {
    switch (op) {
        case B_ADD: return a + b;
        case B_SUB: return a - b;
        case B_MUL: return a * b;
        case B_DIV: return b == 0 ? 0 : a / b;
        case B_MAX: return a > b ? a : b;
        case B_MIN: return a < b ? a : b;
        case B_EQUAL: return a == b ? 1 : 0;
        case B_LESS: return a < b ? 1 : 0;
        case B_GREATER: return a > b ? 1 : 0;
        case B_MOD: return b == 0 ? 0 : a % b;
    }
    return a;
}

template <typename T>
__global__ void binary_kernel(int op, const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out,
                              int64_t numel, Bcast bc, int same) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t ia = i, ib = i;
        if (!same) {
            ia = 0;
            ib = 0;
            int64_t rem = i;
            for (int d = bc.rank - 1; d >= 0; --d) {
                const int64_t c = rem % bc.oshape[d];
                rem /= bc.oshape[d];
                ia += c * bc.astride[d];
                ib += c * bc.bstride[d];
            }
        }
        out[i] = binary_apply<T>(op, a[ia], b[ib]);
    }
}

// Fast path for the broadcast patterns generated models actually use (same shape, scalar, bias over the trailing dims,
// per-channel term over NCHW): element i of the output reads operand[(i / inner) % len] (or operand[i] when `full`), so
// a float4 chunk needs no per-element index walk.  Host-side conditions (binary_fast_map) guarantee that a 4-chunk
// never straddles two operand elements (inner % 4 == 0) or is contiguous in the operand (inner == 1, len % 4 == 0).
// Same binary_apply, element by element: bit-identical to binary_kernel.
struct OperandMap {
    unsigned inner, len;
    int full;
};
__device__ __forceinline__ float4 fetch4(const float* __restrict__ p, const OperandMap m, unsigned i) {
    if (m.full) return *reinterpret_cast<const float4*>(p + i);
    if (m.len == 1) {
        const float v = p[0];
        return make_float4(v, v, v, v);
    }
    if (m.inner == 1) return *reinterpret_cast<const float4*>(p + i % m.len);
    const float v = p[(i / m.inner) % m.len];
    return make_float4(v, v, v, v);
}
__device__ __forceinline__ float fetch1(const float* __restrict__ p, const OperandMap m, unsigned i) {
    return m.full ? p[i] : p[(i / m.inner) % m.len];
}
__global__ __launch_bounds__(256) void binary_fast_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                                                          float* __restrict__ out, unsigned n, OperandMap ma, OperandMap mb) {
    const unsigned nvec = n >> 2, gtid = blockIdx.x * 256u + threadIdx.x, gstride = gridDim.x * 256u;
    float4* ov = reinterpret_cast<float4*>(out);
    unsigned v = gtid;
    for (; v + gstride < nvec; v += 2 * gstride) {  // two chunks per trip: both operands' loads in flight together
        const float4 a0 = fetch4(a, ma, 4 * v), b0 = fetch4(b, mb, 4 * v);
        const float4 a1 = fetch4(a, ma, 4 * (v + gstride)), b1 = fetch4(b, mb, 4 * (v + gstride));
        ov[v] = make_float4(binary_apply<float>(op, a0.x, b0.x), binary_apply<float>(op, a0.y, b0.y),
                            binary_apply<float>(op, a0.z, b0.z), binary_apply<float>(op, a0.w, b0.w));
        ov[v + gstride] = make_float4(binary_apply<float>(op, a1.x, b1.x), binary_apply<float>(op, a1.y, b1.y),
                                      binary_apply<float>(op, a1.z, b1.z), binary_apply<float>(op, a1.w, b1.w));
    }
    if (v < nvec) {
        const float4 a0 = fetch4(a, ma, 4 * v), b0 = fetch4(b, mb, 4 * v);
        ov[v] = make_float4(binary_apply<float>(op, a0.x, b0.x), binary_apply<float>(op, a0.y, b0.y),
                            binary_apply<float>(op, a0.z, b0.z), binary_apply<float>(op, a0.w, b0.w));
    }
    for (unsigned i = 4 * nvec + gtid; i < n; i += gstride) out[i] = binary_apply<float>(op, fetch1(a, ma, i), fetch1(b, mb, i));
}

// sqrt(pow(x[.., a0:a0+L, ..], e0) + pow(x[.., b0:b0+L, ..], e1)): the six-node spectrum-magnitude chain Slice, Pow, Slice, Pow,
// Add, Sqrt in one pass.  Same device functions in the same order as the separate kernels (binary_apply B_POW -> powf with the
// exponent as a run-time value, B_ADD, unary U_SQRT), so the result is the same bits.  x is [outer, dim, inner].
__global__ void halves_pow_add_sqrt_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t dim, int64_t a0, int64_t b0, int64_t len,
                                           int64_t inner, int64_t n, float e0, float e1) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t per = len * inner, o = i / per, r = i - o * per;
        const float* p = x + o * dim * inner + r;
        const float a = binary_apply<float>(B_POW, p[a0 * inner], e0), b = binary_apply<float>(B_POW, p[b0 * inner], e1);
        y[i] = unary_apply(U_SQRT, binary_apply<float>(B_ADD, a, b), true);
    }
}

// (a + b) + c element-wise on equal shapes: the two consecutive residual adds of a transformer block in one pass, same
// rounding order as two `add` kernels
__global__ __launch_bounds__(256) void add3_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c,
                                                   float* __restrict__ out, int64_t n) {
    const int64_t gtid = (int64_t)blockIdx.x * 256 + threadIdx.x, gstride = (int64_t)gridDim.x * 256;
    const bool vec = ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)out) & 15) == 0);
    const int64_t nvec = vec ? n >> 2 : 0;
    for (int64_t i = gtid; i < nvec; i += gstride) {
        const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i], z = reinterpret_cast<const float4*>(c)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4((x.x + y.x) + z.x, (x.y + y.y) + z.y, (x.z + y.z) + z.z, (x.w + y.w) + z.w);
    }
    for (int64_t i = 4 * nvec + gtid; i < n; i += gstride) out[i] = (a[i] + b[i]) + c[i];
}

// where_op (manipulation.rs:1215-): out = cond != 0 ? x : y, three-way broadcast
__global__ void where_kernel(const float* __restrict__ cnd, const float* __restrict__ x, const float* __restrict__ y,
                             float* __restrict__ out, int64_t numel, Bcast bc) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t ic = 0, ix = 0, iy = 0, rem = i;
        for (int d = bc.rank - 1; d >= 0; --d) {
            const int64_t c = rem % bc.oshape[d];
            rem /= bc.oshape[d];
            ic += c * bc.cstride[d];
            ix += c * bc.astride[d];
            iy += c * bc.bstride[d];
        }
        out[i] = cnd[ic] != 0.0f ? x[ix] : y[iy];
    }
}

__global__ void clip_kernel(const float* __restrict__ x, float lo, float hi, float* __restrict__ y, int64_t len) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
        float v = x[i];
        v = v < lo ? lo : v;  // f32::clamp
        v = v > hi ? hi : v;
        y[i] = v;
    }
}

// ------------------------------------------------------------------------------------------ reductions
enum ReduceOp { R_SUM = 0, R_MEAN = 1, R_MAX = 2, R_L2 = 3, R_MIN = 4 };
struct ReduceDesc {
    int rank_keep, rank_red;
    int64_t keep_shape[LELE_MAX_RANK], keep_stride[LELE_MAX_RANK];  // input strides of kept dims
    int64_t red_shape[LELE_MAX_RANK], red_stride[LELE_MAX_RANK];    // input strides of reduced dims (row-major order)
    int64_t red_count;
};
// one thread per output element; reduced elements visited in row-major input order (math.rs:1580-1597)
__global__ void reduce_kernel(int op, const float* __restrict__ x, float* __restrict__ out, int64_t out_numel,
                              ReduceDesc rd) {
    const int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= out_numel) return;
    int64_t base = 0, rem = o;
    for (int d = rd.rank_keep - 1; d >= 0; --d) {
        base += (rem % rd.keep_shape[d]) * rd.keep_stride[d];
        rem /= rd.keep_shape[d];
    }
    float acc = (op == R_MAX) ? -INFINITY : (op == R_MIN ? INFINITY : 0.0f);
    for (int64_t r = 0; r < rd.red_count; ++r) {
        int64_t off = base, rr = r;
        for (int d = rd.rank_red - 1; d >= 0; --d) {
            off += (rr % rd.red_shape[d]) * rd.red_stride[d];
            rr /= rd.red_shape[d];
        }
        const float v = x[off];
        if (op == R_MAX)
            acc = v > acc ? v : acc;
        else if (op == R_MIN)
            acc = v < acc ? v : acc;
        else if (op == R_L2)
            acc = acc + v * v;
        else
            acc = acc + v;
    }
    if (op == R_MEAN) acc = acc * (1.0f / (float)rd.red_count);  // math.rs:1602-1605
    if (op == R_L2) acc = sqrtf(acc);
    out[o] = acc;
}

// max / min over a contiguous innermost axis (order-free, so any tree gives the reference's value): 16 lanes per output row read
// the row coalesced and meet by shuffle -- the one-thread-per-output form above reads it with a row-length lane stride
__global__ __launch_bounds__(256) void reduce_minmax_last_kernel(int op, const float* __restrict__ x, float* __restrict__ out, int64_t rows,
                                                                 int64_t n) {
    const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int l = threadIdx.x & 15;
    // the sequential scan keeps the FIRST of equal values (only +0 / -0 can tell): carry the position, earlier wins a tie
    float acc = op == R_MAX ? -INFINITY : INFINITY;
    int64_t at = n;
    if (r < rows) {
        const float* row = x + r * n;
        for (int64_t j = l; j < n; j += 16) {
            const float v = row[j];
            if (op == R_MAX ? v > acc : v < acc) acc = v, at = j;  // NaN never wins, as in `v > acc ? v : acc`
        }
    }
    for (int off = 8; off > 0; off >>= 1) {
        const float o = __shfl_xor(acc, off, 16);
        const int64_t oa = __shfl_xor(at, off, 16);
        if ((op == R_MAX ? o > acc : o < acc) || (o == acc && oa < at)) acc = o, at = oa;
    }
    if (r < rows && l == 0) out[r] = acc;
}

// The same for FEW LONG rows (a global max over activations is one row): with 16 lanes a row the kernel above would scan a whole
// tensor from one workgroup.  Stage 1: grid (pieces, rows), a workgroup reduces its piece of a row to (value, position of the first
// occurrence); stage 2: one wave per row merges the pieces, the earlier position winning a tie -- the sequential scan's value.
__device__ __forceinline__ void minmax_take(int op, float& acc, int64_t& at, float o, int64_t oa) {
    if ((op == R_MAX ? o > acc : o < acc) || (o == acc && oa < at)) acc = o, at = oa;
}
__global__ __launch_bounds__(256) void reduce_minmax_part_kernel(int op, const float* __restrict__ x, float* __restrict__ pv,
                                                                 long long* __restrict__ pa, int64_t n, int64_t per) {
    __shared__ float sv[4];
    __shared__ long long sa[4];
    const float* row = x + (int64_t)blockIdx.y * n;
    const int64_t c0 = (int64_t)blockIdx.x * per, c1 = c0 + per < n ? c0 + per : n;
    float acc = op == R_MAX ? -INFINITY : INFINITY;
    int64_t at = n;
    for (int64_t j = c0 + threadIdx.x; j < c1; j += 256) {
        const float v = row[j];
        if (op == R_MAX ? v > acc : v < acc) acc = v, at = j;
    }
    for (int off = 32; off > 0; off >>= 1) minmax_take(op, acc, at, __shfl_xor(acc, off), __shfl_xor(at, off));
    if ((threadIdx.x & 63) == 0) sv[threadIdx.x >> 6] = acc, sa[threadIdx.x >> 6] = at;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) minmax_take(op, acc, at, sv[w], (int64_t)sa[w]);
        pv[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = acc;
        pa[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = at;
    }
}
__global__ __launch_bounds__(64) void reduce_minmax_merge_kernel(int op, const float* __restrict__ pv, const long long* __restrict__ pa,
                                                                 float* __restrict__ out, int pieces, int64_t n) {
    float acc = op == R_MAX ? -INFINITY : INFINITY;
    int64_t at = n;
    for (int p = threadIdx.x; p < pieces; p += 64) minmax_take(op, acc, at, pv[(int64_t)blockIdx.x * pieces + p], (int64_t)pa[(int64_t)blockIdx.x * pieces + p]);
    for (int off = 32; off > 0; off >>= 1) minmax_take(op, acc, at, __shfl_xor(acc, off), __shfl_xor(at, off));
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

// ------------------------------------------------------------------------------------------ row statistics
// 32 lanes play the 4x8 accumulator slots of the AVX2 code: slot l = 8*u + i accumulates elements j = 32c + l.
// Returns (in every lane of the 32-lane group) hsum( (s0+s1)+(s2+s3) [+ 8-wide remainder chunks] ) + scalar tail.
template <bool SQUARE, bool PLAIN, class F>
__device__ __forceinline__ void row_sums(const F& elem, int64_t n, int l, float* out_sum, float* out_sq) {
    float s = 0.0f, q = 0.0f;
    int64_t j = 0;
    for (; j + 32 <= n; j += 32) {
        const float v = elem(j + l);
        if (PLAIN) s = s + v;
        if (SQUARE) q = fmaf_(v, v, q);
    }
    // merge (acc0+acc1) + (acc2+acc3): slot i of the result lives in lanes 0..7
    float s01 = s + __shfl_down(s, 8, 32), q01 = q + __shfl_down(q, 8, 32);
    float sv = s01 + __shfl_down(s01, 16, 32), qv = q01 + __shfl_down(q01, 16, 32);
    for (; j + 8 <= n; j += 8) {  // remaining 8-wide chunks go to the merged vector
        if (l < 8) {
            const float v = elem(j + l);
            if (PLAIN) sv = sv + v;
            if (SQUARE) qv = fmaf_(v, v, qv);
        }
    }
    // horizontal: (i)+(i+4), then (i)+(i+2), then [0]+[1]
    sv = sv + __shfl_down(sv, 4, 32);
    qv = qv + __shfl_down(qv, 4, 32);
    sv = sv + __shfl_down(sv, 2, 32);
    qv = qv + __shfl_down(qv, 2, 32);
    sv = sv + __shfl_down(sv, 1, 32);
    qv = qv + __shfl_down(qv, 1, 32);
    for (; j < n; ++j) {  // scalar tail (lane 0 carries the result)
        const float v = elem(j);
        if (PLAIN) sv = sv + v;
        if (SQUARE) qv = qv + v * v;
    }
    *out_sum = __shfl(sv, 0, 32);
    *out_sq = __shfl(qv, 0, 32);
}

// One trip to memory per element: the row lives in registers between the statistics and the output pass.
// RPB rows per block (32 lanes each); NT = ceil(norm / 32) rounded up to the instantiated sizes.
template <int NT>
__global__ void layer_norm_reg_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                      const float* __restrict__ b, float* __restrict__ y, int norm, int64_t outer,
                                      float eps, float* __restrict__ rowstat /* NULL or [outer][2]: min, max of each output row */) {
    const int l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (row >= outer) return;
    const float* in = x + row * norm;
    float* out = y + row * norm;
    float v[NT], gg[NT], bb[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = 32 * c + l, jc = j < norm ? j : norm - 1;  // clamped, unconditional: all loads in flight together
        v[c] = in[jc];
        gg[c] = g[jc];
        bb[c] = b[jc];
    }
    float sum, sumsq;
    row_sums_reg<NT, true, true>(v, norm, l, &sum, &sumsq);
    const float inv_n = 1.0f / (float)norm;
    const float mean = sum * inv_n;
    const float var = sumsq * inv_n - mean * mean;
    const float inv_std = 1.0f / sqrtf(var + eps);
    const int body = norm & ~7;
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = 32 * c + l;
        if (j < norm) {
            const float t = (v[c] - mean) * inv_std;
            const float o = j < body ? fmaf_(t, gg[c], bb[c]) : t * gg[c] + bb[c];
            out[j] = o;
            mn = o < mn ? o : mn;  // the comparisons of qminmax_kernel (quant.hip): min / max are order-independent, exact
            mx = o > mx ? o : mx;
        }
    }
    if (rowstat) {
        mn = group_allreduce32(mn, [](float cur, float a) { return a < cur ? a : cur; });
        mx = group_allreduce32(mx, [](float cur, float a) { return a > cur ? a : cur; });
        if (l == 0) {
            rowstat[2 * row] = mn;
            rowstat[2 * row + 1] = mx;
        }
    }
}

// `scale` (NULL or one f32 on the device): softmax(x * scale[0]) with the product rounded to f32 first, i.e. exactly what
// a `mul` kernel followed by this kernel computes (lele_hip_softmax_scaled)
template <int NT>
__global__ void softmax_reg_kernel(const float* __restrict__ x, float* __restrict__ y, int len, int64_t outer,
                                   const float* __restrict__ scale) {
    const int l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (row >= outer) return;
    const float* src = x + row * len;
    float* dst = y + row * len;
    const bool scaled = scale != nullptr;
    const float sc = scaled ? scale[0] : 1.0f;
    float v[NT];
    float m = -3.40282347e+38f;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = 32 * c + l;
        v[c] = src[j < len ? j : len - 1];
        if (scaled) v[c] = v[c] * sc;
        if (j < len) m = fmaxf(m, v[c]);
    }
    m = group_max32(m);
    const int body = len & ~7;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = 32 * c + l;
        // computed once, reused for the sum and the output; register rows entirely inside the 8-wide body skip the libm call
        // (uniform condition -- a per-lane select would evaluate both functions for every element)
        if (32 * c + 32 <= body) v[c] = exp_poly(v[c] - m);
        else v[c] = j < body ? exp_poly(v[c] - m) : expf(v[c] - m);
    }
    float sum, dummy;
    row_sums_reg<NT, false, true>(v, len, l, &sum, &dummy);
    const float inv_sum = 1.0f / sum;
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int j = 32 * c + l;
        if (j < len) dst[j] = v[c] * inv_sum;
    }
}

// layer_norm_x86, avx/norm.rs:10-133.  8 rows per 256-thread block (one 32-lane group per row).
__global__ __launch_bounds__(256) void layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                         const float* __restrict__ b, float* __restrict__ y,
                                                         int64_t norm, int64_t outer, float eps) {
    const int l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= outer) return;
    const float* in = x + row * norm;
    float* out = y + row * norm;
    float sum, sumsq;
    row_sums<true, true>([&](int64_t j) { return in[j]; }, norm, l, &sum, &sumsq);
    const float inv_n = 1.0f / (float)norm;
    const float mean = sum * inv_n;
    const float var = sumsq * inv_n - mean * mean;
    const float inv_std = 1.0f / sqrtf(var + eps);
    const int64_t body = norm & ~int64_t(7);
    for (int64_t j = l; j < norm; j += 32) {
        const float t = (in[j] - mean) * inv_std;
        out[j] = j < body ? fmaf_(t, g[j], b[j]) : t * g[j] + b[j];
    }
}

// rms_norm_x86, avx/norm.rs:236-307
__global__ __launch_bounds__(256) void rms_norm_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                       float* __restrict__ y, int64_t norm, int64_t outer, float eps) {
    const int l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= outer) return;
    const float* in = x + row * norm;
    float* out = y + row * norm;
    float dummy, sumsq;
    row_sums<true, false>([&](int64_t j) { return in[j]; }, norm, l, &dummy, &sumsq);
    const float inv_n = 1.0f / (float)norm;
    const float rms_inv = 1.0f / sqrtf(sumsq * inv_n + eps);
    const int64_t body = norm & ~int64_t(7);
    for (int64_t j = l; j < norm; j += 32)
        out[j] = j < body ? in[j] * (w[j] * rms_inv) : in[j] * rms_inv * w[j];  // body: eff_w = weight*rms_inv
}

// softmax over a contiguous last axis, avx/norm.rs:139-229
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int64_t len,
                                                      int64_t outer, const float* __restrict__ scale) {
    const int l = threadIdx.x & 31;
    const int64_t row = (int64_t)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= outer) return;
    const float* src = x + row * len;
    float* dst = y + row * len;
    const bool scaled = scale != nullptr;
    const float sc = scaled ? scale[0] : 1.0f;
    auto in = [&](int64_t j) { return scaled ? src[j] * sc : src[j]; };
    float m = -3.40282347e+38f;  // f32::MIN seeds; max is order-independent
    for (int64_t j = l; j < len; j += 32) m = fmaxf(m, in(j));
    m = group_max32(m);
    const int64_t body = len & ~int64_t(7);
    // exp(x - max): polynomial in the SIMD body, libm in the tail; summed in the AVX accumulator order.  The values
    // are recomputed (deterministically) wherever another lane's element is needed, so no cross-lane memory traffic.
    auto ev = [&](int64_t j) { return j < body ? exp_poly(in(j) - m) : expf(in(j) - m); };
    float sum, dummy;
    row_sums<false, true>(ev, len, l, &sum, &dummy);
    const float inv_sum = 1.0f / sum;
    for (int64_t j = l; j < len; j += 32) dst[j] = ev(j) * inv_sum;
}

// batch_norm (norm.rs:313-418): per (outer, channel) slice out = fma(x, scale_val, bias_val) in the 8-wide body
__global__ void batch_norm_kernel(const float* __restrict__ x, const float* __restrict__ s, const float* __restrict__ b,
                                  const float* __restrict__ m, const float* __restrict__ v, float eps, int64_t c,
                                  int64_t inner, int64_t numel, float* __restrict__ y) {
    const int64_t body = inner & ~int64_t(7);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = i % inner, ch = (i / inner) % c;
        const float scale_val = s[ch] / sqrtf(v[ch] + eps);
        const float bias_val = b[ch] - m[ch] * scale_val;
        y[i] = k < body ? fmaf_(x[i], scale_val, bias_val) : x[i] * scale_val + bias_val;
    }
}

inline int grid_for(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 4096)); }

// numpy-style broadcast of up to three operands (utils.rs:103-132)
int make_bcast(const LeleTensor* const* ops, int nops, Bcast* bc, std::vector<int64_t>* oshape, const char* who) {
    int rank = 0;
    for (int i = 0; i < nops; ++i) rank = std::max(rank, (int)ops[i]->rank);
    LELE_REQUIRE(rank <= LELE_MAX_RANK, "%s: rank %d exceeds %d", who, rank, LELE_MAX_RANK);
    oshape->assign(rank, 1);
    for (int d = 0; d < rank; ++d)
        for (int i = 0; i < nops; ++i) {
            const int od = d - (rank - ops[i]->rank);
            const int64_t dim = od < 0 ? 1 : ops[i]->shape[od];
            if (dim != 1) {
                LELE_REQUIRE((*oshape)[d] == 1 || (*oshape)[d] == dim, "Shapes not broadcastable");  // math.rs:80-85
                (*oshape)[d] = dim;
            }
        }
    for (int d = 0; d < rank; ++d) {
        const int od = d - (rank - ops[0]->rank);
        if (od >= 0 && ops[0]->shape[od] == 0) (*oshape)[d] = 0;
    }
    bc->rank = rank;
    int64_t* strides[3] = {bc->astride, bc->bstride, bc->cstride};
    for (int i = 0; i < nops; ++i) {
        int64_t st = 1;
        for (int d = rank - 1; d >= 0; --d) {
            const int od = d - (rank - ops[i]->rank);
            const int64_t dim = od < 0 ? 1 : ops[i]->shape[od];
            strides[i][d] = dim == 1 ? 0 : st;
            st *= dim;
        }
    }
    for (int d = 0; d < rank; ++d) bc->oshape[d] = (*oshape)[d];
    return 0;
}

// OperandMap of one operand of a broadcast (strides from make_bcast), or false when the pattern is not a single
// contiguous run of non-broadcast dimensions / not float4-friendly.
bool binary_fast_map(const Bcast& bc, const int64_t* stride, const void* ptr, int64_t n, int64_t count, OperandMap* m) {
    if (((uintptr_t)ptr & 15) != 0) return false;
    if (count == n) {
        *m = OperandMap{1u, 1u, 1};
        return true;
    }
    int first = -1, last = -1;
    for (int d = 0; d < bc.rank; ++d)
        if (stride[d] != 0 && bc.oshape[d] != 1) {
            if (first < 0) first = d;
            last = d;
        }
    if (first < 0) {  // a single element
        *m = OperandMap{1u, 1u, 0};
        return true;
    }
    for (int d = first; d <= last; ++d)
        if (stride[d] == 0 && bc.oshape[d] != 1) return false;  // a broadcast dimension inside the run
    int64_t inner = 1, len = 1;
    for (int d = last + 1; d < bc.rank; ++d) inner *= bc.oshape[d];
    for (int d = first; d <= last; ++d) len *= bc.oshape[d];
    if (!(inner % 4 == 0 || (inner == 1 && len % 4 == 0))) return false;
    *m = OperandMap{(unsigned)inner, (unsigned)len, 0};
    return true;
}

// Same-shape binary op whose operands and / or result are CHANNEL VIEWS of wider NCHW tensors (LelePitch, lele_hip.h): image n of
// an operand starts n * pitch elements after image 0 and is dense inside.  binary_apply element by element: the bits of
// lele_hip_binary on dense copies.  grid (chunks of one image, images).
__global__ __launch_bounds__(256) void binary_pitched_kernel(int op, const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ out, unsigned per_image, long long pa, long long pb,
                                                             long long po, int vec) {
    const float* ai = a + (long long)blockIdx.y * pa;
    const float* bi = b + (long long)blockIdx.y * pb;
    float* oi = out + (long long)blockIdx.y * po;
    const unsigned gtid = blockIdx.x * 256u + threadIdx.x, gstride = gridDim.x * 256u;
    if (vec) {
        const unsigned nvec = per_image >> 2;
        for (unsigned v = gtid; v < nvec; v += gstride) {
            const float4 x = reinterpret_cast<const float4*>(ai)[v], y = reinterpret_cast<const float4*>(bi)[v];
            reinterpret_cast<float4*>(oi)[v] = make_float4(binary_apply<float>(op, x.x, y.x), binary_apply<float>(op, x.y, y.y),
                                                           binary_apply<float>(op, x.z, y.z), binary_apply<float>(op, x.w, y.w));
        }
        for (unsigned i = 4 * nvec + gtid; i < per_image; i += gstride) oi[i] = binary_apply<float>(op, ai[i], bi[i]);
    } else {
        for (unsigned i = gtid; i < per_image; i += gstride) oi[i] = binary_apply<float>(op, ai[i], bi[i]);
    }
}

}  // namespace

extern "C" {

int lele_hip_unary(LeleCtx* ctx, int op, const LeleTensor* x, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "unary: NULL argument");
    LELE_REQUIRE(op >= 0 && op <= U_CEIL, "unary: unknown op %d", op);
    LELE_REQUIRE(x->dtype == LELE_F32, "unary: f32 input required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t len = numel(x);
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)len * 4));
    if (len) {
        if ((((uintptr_t)dx | (uintptr_t)out->data) & 15) == 0)
            hipLaunchKernelGGL(unary_vec4_kernel, dim3(grid_for((len + 3) / 4)), dim3(256), 0, ctx->stream, op,
                               (const float*)dx, (float*)out->data, len);
        else
            hipLaunchKernelGGL(unary_kernel, dim3(grid_for(len)), dim3(256), 0, ctx->stream, op, (const float*)dx,
                               (float*)out->data, len);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

int lele_hip_binary(LeleCtx* ctx, int op, const LeleTensor* a, const LeleTensor* b, LeleBuf* out, int64_t* out_shape,
                    int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && b && out, "binary: NULL argument");
    LELE_REQUIRE(op >= 0 && op <= B_OR, "binary: unknown op %d", op);
    LELE_REQUIRE(a->dtype == b->dtype && (a->dtype == LELE_F32 || a->dtype == LELE_I64),
                 "binary: operands must both be f32 or both i64");
    LELE_REQUIRE(a->dtype == LELE_F32 || (op != B_POW && op != B_PRELU && op != B_AND && op != B_OR),
                 "binary: op %d is f32-only", op);
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    Bcast bc;
    std::vector<int64_t> oshape;
    const LeleTensor* ops[2] = {a, b};
    LELE_TRY(make_bcast(ops, 2, &bc, &oshape, "binary"));
    int64_t n = 1;
    for (int64_t d : oshape) n *= d;
    const size_t es = dtype_size(a->dtype);
    LELE_TRY(ctx->arena_reset());
    const void *da = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    LELE_TRY(out->reserve((size_t)n * es));
    if (n) {
        const int same = (numel(a) == n && numel(b) == n) ? 1 : 0;
        OperandMap ma, mb;
        if (a->dtype == LELE_F32 && n < (int64_t(1) << 32) && (((uintptr_t)out->data) & 15) == 0 &&
            binary_fast_map(bc, bc.astride, da, n, numel(a), &ma) && binary_fast_map(bc, bc.bstride, db, n, numel(b), &mb))
            hipLaunchKernelGGL(binary_fast_kernel, dim3(grid_for((n + 7) / 8)), dim3(256), 0, ctx->stream, op, (const float*)da,
                               (const float*)db, (float*)out->data, (unsigned)n, ma, mb);
        else if (a->dtype == LELE_F32)
            hipLaunchKernelGGL(binary_kernel<float>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, op, (const float*)da,
                               (const float*)db, (float*)out->data, n, bc, same);
        else
            hipLaunchKernelGGL(binary_kernel<int64_t>, dim3(grid_for(n)), dim3(256), 0, ctx->stream, op,
                               (const int64_t*)da, (const int64_t*)db, (int64_t*)out->data, n, bc, same);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

int lele_hip_binary_pitched(LeleCtx* ctx, int op, const LeleTensor* a, const LeleTensor* b, const LelePitch* pitch, LeleBuf* out,
                            int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && b && out && pitch, "binary_pitched: NULL argument");
    LELE_REQUIRE(op >= 0 && op <= B_OR, "binary: unknown op %d", op);
    LELE_REQUIRE(a->dtype == LELE_F32 && b->dtype == LELE_F32, "binary_pitched: f32 operands required");
    LELE_REQUIRE(a->rank == b->rank && a->rank >= 2, "binary_pitched: operands of the same shape (rank >= 2) required");
    for (int i = 0; i < a->rank; ++i) LELE_REQUIRE(a->shape[i] == b->shape[i], "binary_pitched: operands of the same shape required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t images = a->shape[0], per = images ? numel(a) / images : 0;
    LELE_REQUIRE(per < (int64_t(1) << 32) && images <= 65535, "binary_pitched: image too large for the 32-bit kernel");
    LELE_REQUIRE((pitch->x_pitch == 0 || (a->mem == LELE_MEM_DEVICE && pitch->x_pitch >= per)) &&
                 (pitch->y_pitch == 0 || (b->mem == LELE_MEM_DEVICE && pitch->y_pitch >= per)),
                 "binary_pitched: a pitch needs a device tensor and must cover one image");
    LELE_TRY(ctx->arena_reset());
    const void *da = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    float* dst = nullptr;
    LELE_TRY(lele::pitched_out(out, pitch, images, per, 4, (void**)&dst));
    if (images * per) {
        const long long pa = pitch->x_pitch ? pitch->x_pitch : per, pb = pitch->y_pitch ? pitch->y_pitch : per,
                        po = pitch->out_pitch ? pitch->out_pitch : per;
        const int vec = ((((uintptr_t)da) | ((uintptr_t)db) | ((uintptr_t)dst)) & 15) == 0 && pa % 4 == 0 && pb % 4 == 0 && po % 4 == 0;
        const unsigned chunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>((per + 2047) / 2048, 4096));
        hipLaunchKernelGGL(binary_pitched_kernel, dim3(chunks, (unsigned)images), dim3(256), 0, ctx->stream, op, (const float*)da,
                           (const float*)db, dst, (unsigned)per, pa, pb, po, vec);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(a->shape, a->shape + a->rank));
}

int lele_hip_where(LeleCtx* ctx, const LeleTensor* cond, const LeleTensor* x, const LeleTensor* y, LeleBuf* out,
                   int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && cond && x && y && out, "where_op: NULL argument");
    LELE_REQUIRE(cond->dtype == LELE_F32 && x->dtype == LELE_F32 && y->dtype == LELE_F32,
                 "where_op: condition and values must be f32 tensors");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    Bcast bc;
    std::vector<int64_t> oshape;
    const LeleTensor* ops[3] = {x, y, cond};
    LELE_TRY(make_bcast(ops, 3, &bc, &oshape, "where_op"));
    int64_t n = 1;
    for (int64_t d : oshape) n *= d;
    LELE_TRY(ctx->arena_reset());
    const void *dc = nullptr, *dx = nullptr, *dy = nullptr;
    LELE_TRY(ctx->dev_ptr(cond, &dc));
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(y, &dy));
    LELE_TRY(out->reserve((size_t)n * 4));
    if (n) {
        hipLaunchKernelGGL(where_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const float*)dc,
                           (const float*)dx, (const float*)dy, (float*)out->data, n, bc);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

int lele_hip_clip(LeleCtx* ctx, const LeleTensor* x, int has_min, float min_v, int has_max, float max_v, LeleBuf* out,
                  int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "clip: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t len = numel(x);
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)len * 4));
    if (len) {
        hipLaunchKernelGGL(clip_kernel, dim3(grid_for(len)), dim3(256), 0, ctx->stream, (const float*)dx,
                           has_min ? min_v : -3.40282347e+38f, has_max ? max_v : 3.40282347e+38f, (float*)out->data,
                           len);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

int lele_hip_reduce(LeleCtx* ctx, int op, const LeleTensor* x, const int64_t* axes, size_t naxes, int keepdims,
                    LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "reduce: NULL argument");
    LELE_REQUIRE(op >= 0 && op <= R_MIN, "reduce: unknown op %d", op);
    LELE_REQUIRE(x->rank <= LELE_MAX_RANK, "reduce: rank too large");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int dims = x->rank;
    std::vector<bool> mask(dims, false);
    for (size_t i = 0; i < naxes; ++i) {  // math.rs:1534-1551
        int64_t ax = axes[i] < 0 ? dims + axes[i] : axes[i];
        LELE_REQUIRE(ax >= 0 && ax < dims, "reduce: axis %lld out of range", (long long)axes[i]);
        mask[ax] = true;
    }
    if (naxes == 0)
        for (int d = 0; d < dims; ++d) mask[d] = true;  // ONNX default: reduce all
    std::vector<int64_t> istr(dims, 1);
    for (int d = dims - 2; d >= 0; --d) istr[d] = istr[d + 1] * x->shape[d + 1];
    ReduceDesc rd{};
    std::vector<int64_t> oshape;
    rd.red_count = 1;
    for (int d = 0; d < dims; ++d) {
        if (mask[d]) {
            rd.red_shape[rd.rank_red] = x->shape[d];
            rd.red_stride[rd.rank_red++] = istr[d];
            rd.red_count *= x->shape[d];
            if (keepdims) oshape.push_back(1);
        } else {
            rd.keep_shape[rd.rank_keep] = x->shape[d];
            rd.keep_stride[rd.rank_keep++] = istr[d];
            oshape.push_back(x->shape[d]);
        }
    }
    int64_t on = 1;
    for (int64_t d : oshape) on *= d;
    LELE_TRY(ctx->arena_reset());
    const void* dx = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)on * 4));
    if (on) {
        bool rows_first = (op == R_MAX || op == R_MIN) && rd.rank_red == 1 && rd.red_stride[0] == 1 && rd.red_count >= 16 && on < (int64_t(1) << 34);
        int64_t expect = rd.red_count;  // the kept dims must be laid out row after row: output o starts at o * red_count
        for (int d = rd.rank_keep - 1; rows_first && d >= 0; --d) {
            rows_first = rd.keep_stride[d] == expect;
            expect *= rd.keep_shape[d];
        }
        if (rows_first && on <= 2 * (int64_t)ctx->num_cus && rd.red_count >= 65536 && on <= 65535) {  // few long rows: two stages
            const int64_t per = std::max<int64_t>(8192, (rd.red_count + 1023) / 1024);
            const int pieces = (int)((rd.red_count + per - 1) / per);
            void *pv = nullptr, *pa = nullptr;
            LELE_TRY(ctx->arena_alloc((size_t)on * pieces * 4, &pv));
            LELE_TRY(ctx->arena_alloc((size_t)on * pieces * 8, &pa));
            hipLaunchKernelGGL(reduce_minmax_part_kernel, dim3((unsigned)pieces, (unsigned)on), dim3(256), 0, ctx->stream, op, (const float*)dx,
                               (float*)pv, (long long*)pa, rd.red_count, per);
            hipLaunchKernelGGL(reduce_minmax_merge_kernel, dim3((unsigned)on), dim3(64), 0, ctx->stream, op, (const float*)pv, (const long long*)pa,
                               (float*)out->data, pieces, rd.red_count);
        } else if (rows_first)
            hipLaunchKernelGGL(reduce_minmax_last_kernel, dim3((unsigned)((on + 15) / 16)), dim3(256), 0, ctx->stream, op, (const float*)dx,
                               (float*)out->data, on, rd.red_count);
        else
            hipLaunchKernelGGL(reduce_kernel, dim3((unsigned)((on + 63) / 64)), dim3(64), 0, ctx->stream, op,
                               (const float*)dx, (float*)out->data, on, rd);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, oshape);
}

int lele_hip_layer_norm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, const LeleTensor* bias,
                        int32_t axis, float epsilon, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && scale && bias && out, "layer_norm: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int nd = x->rank;
    const int ax = axis < 0 ? nd + axis : axis;  // norm.rs:234-235
    LELE_REQUIRE(ax >= 0 && ax <= nd, "layer_norm: axis %d out of range", axis);
    int64_t outer = 1, norm = 1;
    for (int d = 0; d < ax; ++d) outer *= x->shape[d];
    for (int d = ax; d < nd; ++d) norm *= x->shape[d];
    LELE_REQUIRE(numel(scale) >= norm && numel(bias) >= norm, "layer_norm: scale/bias shorter than the normalised size");
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dg = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(scale, &dg));
    LELE_TRY(ctx->dev_ptr(bias, &db));
    LELE_TRY(out->reserve((size_t)outer * norm * 4));
    if (outer * norm) {
        // rows that fit 32 registers per lane take the single-pass kernel; few rows -> fewer rows per block (more CUs)
        const int rpb = outer >= 4096 ? 8 : (outer >= 1024 ? 4 : 2);
        const dim3 rgrid((unsigned)((outer + rpb - 1) / rpb)), rblock(32 * rpb);
        // row statistics for a dynamic quantisation that may read this result next (common.h, LeleBuf::rowstat)
        float* rs = nullptr;
        if (norm <= 1024) {
            LELE_TRY(out->reserve_rowstat(outer));
            if ((size_t)outer <= out->rowstat_cap) rs = out->rowstat;
        }
        if (norm <= 256)
            hipLaunchKernelGGL(layer_norm_reg_kernel<8>, rgrid, rblock, 0, ctx->stream, (const float*)dx, (const float*)dg,
                               (const float*)db, (float*)out->data, (int)norm, outer, epsilon, rs);
        else if (norm <= 512)
            hipLaunchKernelGGL(layer_norm_reg_kernel<16>, rgrid, rblock, 0, ctx->stream, (const float*)dx, (const float*)dg,
                               (const float*)db, (float*)out->data, (int)norm, outer, epsilon, rs);
        else if (norm <= 1024)
            hipLaunchKernelGGL(layer_norm_reg_kernel<32>, rgrid, rblock, 0, ctx->stream, (const float*)dx, (const float*)dg,
                               (const float*)db, (float*)out->data, (int)norm, outer, epsilon, rs);
        else
            hipLaunchKernelGGL(layer_norm_kernel, dim3((unsigned)((outer + 7) / 8)), dim3(256), 0, ctx->stream,
                               (const float*)dx, (const float*)dg, (const float*)db, (float*)out->data, norm, outer, epsilon);
        LELE_HIP_CHECK(hipGetLastError());
        if (rs) {
            out->rowstat_rows = outer;
            out->rowstat_len = norm;
            out->rowstat_kind = 0;
            out->rowstat_valid = true;
        }
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

int lele_hip_rms_norm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* weight, int32_t axis, float epsilon,
                      LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && weight && out, "rms_norm: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int nd = x->rank;
    const int ax = axis < 0 ? nd + axis : axis;
    LELE_REQUIRE(ax >= 0 && ax <= nd, "rms_norm: axis %d out of range", axis);
    int64_t outer = 1, norm = 1;
    for (int d = 0; d < ax; ++d) outer *= x->shape[d];
    for (int d = ax; d < nd; ++d) norm *= x->shape[d];
    LELE_REQUIRE(numel(weight) >= norm, "rms_norm: weight shorter than the normalised size");
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dw = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(weight, &dw));
    LELE_TRY(out->reserve((size_t)outer * norm * 4));
    if (outer * norm) {
        hipLaunchKernelGGL(rms_norm_kernel, dim3((unsigned)((outer + 7) / 8)), dim3(256), 0, ctx->stream,
                           (const float*)dx, (const float*)dw, (float*)out->data, norm, outer, epsilon);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

static int softmax_impl(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, int32_t axis, LeleBuf* out, int64_t* out_shape,
                        int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && out, "softmax: NULL argument");
    LELE_REQUIRE(!scale || (scale->dtype == LELE_F32 && numel(scale) == 1), "softmax_scaled: the scale must be one f32 value");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int nd = x->rank;
    const int ax = axis < 0 ? nd + axis : axis;
    LELE_REQUIRE(ax >= 0 && ax < nd, "softmax: axis %d out of range", axis);  // norm.rs:15
    int64_t inner = 1, outer = 1;
    for (int d = ax + 1; d < nd; ++d) inner *= x->shape[d];
    for (int d = 0; d < ax; ++d) outer *= x->shape[d];
    LELE_REQUIRE(inner == 1, "Softmax only supported on last dimension for now");  // norm.rs:218
    const int64_t len = x->shape[ax];
    LELE_TRY(ctx->arena_reset());
    const void *dx = nullptr, *dsc = nullptr;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    if (scale) LELE_TRY(ctx->dev_ptr(scale, &dsc));
    LELE_TRY(out->reserve((size_t)outer * len * 4));
    if (outer * len) {
        const int rpb = outer >= 4096 ? 8 : (outer >= 1024 ? 4 : 2);
        const dim3 rgrid((unsigned)((outer + rpb - 1) / rpb)), rblock(32 * rpb);
        if (len <= 256)
            hipLaunchKernelGGL(softmax_reg_kernel<8>, rgrid, rblock, 0, ctx->stream, (const float*)dx, (float*)out->data,
                               (int)len, outer, (const float*)dsc);
        else if (len <= 512)
            hipLaunchKernelGGL(softmax_reg_kernel<16>, rgrid, rblock, 0, ctx->stream, (const float*)dx, (float*)out->data,
                               (int)len, outer, (const float*)dsc);
        else if (len <= 1024)
            hipLaunchKernelGGL(softmax_reg_kernel<32>, rgrid, rblock, 0, ctx->stream, (const float*)dx, (float*)out->data,
                               (int)len, outer, (const float*)dsc);
        else
            hipLaunchKernelGGL(softmax_kernel, dim3((unsigned)((outer + 7) / 8)), dim3(256), 0, ctx->stream,
                               (const float*)dx, (float*)out->data, len, outer, (const float*)dsc);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

int lele_hip_softmax(LeleCtx* ctx, const LeleTensor* x, int32_t axis, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    return softmax_impl(ctx, x, nullptr, axis, out, out_shape, out_rank);
}

int lele_hip_softmax_scaled(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, int32_t axis, LeleBuf* out,
                            int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(scale, "softmax_scaled: NULL scale");
    return softmax_impl(ctx, x, scale, axis, out, out_shape, out_rank);
}

int lele_hip_add3(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, const LeleTensor* c, LeleBuf* out, int64_t* out_shape,
                  int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && b && c && out, "add3: NULL argument");
    LELE_REQUIRE(a->dtype == LELE_F32 && b->dtype == LELE_F32 && c->dtype == LELE_F32, "add3: f32 operands required");
    bool same = a->rank == b->rank && a->rank == c->rank;
    for (int d = 0; same && d < a->rank; ++d) same = a->shape[d] == b->shape[d] && a->shape[d] == c->shape[d];
    if (!same) {  // broadcasting operands: the two `add`s this op stands for, the first into a library-owned temporary (the third
                  // operand may broadcast OUTWARD, so the second pass cannot run in place)
        int64_t sh1[LELE_MAX_RANK];
        int32_t r1 = 0;
        LeleBuf* tmp = nullptr;
        LELE_TRY(ctx->tmp_buf(0, &tmp));
        LELE_TRY(lele_hip_binary(ctx, B_ADD, a, b, tmp, sh1, &r1));
        LeleTensor t{tmp->data, sh1, r1, LELE_F32, LELE_MEM_DEVICE};
        return lele_hip_binary(ctx, B_ADD, &t, c, out, out_shape, out_rank);
    }
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t n = numel(a);
    LELE_TRY(ctx->arena_reset());
    const void *da, *db, *dc;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    LELE_TRY(ctx->dev_ptr(c, &dc));
    LELE_TRY(out->reserve((size_t)n * 4));
    if (n) {
        hipLaunchKernelGGL(add3_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, ctx->stream, (const float*)da, (const float*)db,
                           (const float*)dc, (float*)out->data, n);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(a->shape, a->shape + a->rank));
}

int lele_hip_halves_pow_add_sqrt(LeleCtx* ctx, const LeleTensor* x, int32_t axis, int64_t lo_start, int64_t lo_end, int64_t hi_start,
                                 int64_t hi_end, const LeleTensor* exp_lo, const LeleTensor* exp_hi, LeleBuf* out, int64_t* out_shape,
                                 int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && exp_lo && exp_hi && out, "halves_pow_add_sqrt: NULL argument");
    LELE_REQUIRE(x->dtype == LELE_F32 && exp_lo->dtype == LELE_F32 && exp_hi->dtype == LELE_F32, "halves_pow_add_sqrt: f32 operands required");
    LELE_REQUIRE(numel(exp_lo) == 1 && numel(exp_hi) == 1, "halves_pow_add_sqrt: the exponents are one-element tensors");
    LELE_REQUIRE(exp_lo->mem != LELE_MEM_DEVICE && exp_hi->mem != LELE_MEM_DEVICE, "halves_pow_add_sqrt: the exponents are host constants");
    const int ax = axis < 0 ? axis + x->rank : axis;
    LELE_REQUIRE(ax >= 0 && ax < x->rank, "halves_pow_add_sqrt: axis %d out of range for rank %d", axis, x->rank);
    const int64_t dim = x->shape[ax];
    auto clampi = [dim](int64_t v) {  // Slice with step 1: negative counts from the end, then clamp to [0, dim] (manipulation.rs:240-262)
        if (v < 0) v += dim;
        return v < 0 ? 0 : v > dim ? dim : v;
    };
    const int64_t a0 = clampi(lo_start), a1 = clampi(lo_end), b0 = clampi(hi_start), b1 = clampi(hi_end);
    const int64_t len = a1 > a0 ? a1 - a0 : 0;
    LELE_REQUIRE((b1 > b0 ? b1 - b0 : 0) == len, "halves_pow_add_sqrt: the two ranges [%lld, %lld) and [%lld, %lld) differ in length",
                 (long long)a0, (long long)a1, (long long)b0, (long long)b1);
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    int64_t outer = 1, inner = 1;
    for (int d = 0; d < ax; ++d) outer *= x->shape[d];
    for (int d = ax + 1; d < x->rank; ++d) inner *= x->shape[d];
    std::vector<int64_t> shp(x->shape, x->shape + x->rank);
    shp[ax] = len;
    const int64_t n = outer * len * inner;
    LELE_TRY(ctx->arena_reset());
    const void* dx;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(out->reserve((size_t)n * 4));
    if (n) {
        hipLaunchKernelGGL(halves_pow_add_sqrt_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const float*)dx, (float*)out->data,
                           dim, a0, b0, len, inner, n, *(const float*)exp_lo->data, *(const float*)exp_hi->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, shp);
}

int lele_hip_batch_norm(LeleCtx* ctx, const LeleTensor* x, const LeleTensor* scale, const LeleTensor* bias,
                        const LeleTensor* mean, const LeleTensor* var, float epsilon, LeleBuf* out, int64_t* out_shape,
                        int32_t* out_rank) {
    LELE_REQUIRE(ctx && x && scale && bias && mean && var && out, "batch_norm: NULL argument");
    LELE_REQUIRE(x->rank >= 1, "batch_norm: rank >= 1 required");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t c = x->rank > 1 ? x->shape[1] : x->shape[0];  // norm.rs:370
    int64_t inner = 1;
    for (int d = 2; d < x->rank; ++d) inner *= x->shape[d];
    const int64_t n = numel(x);
    LELE_REQUIRE(numel(scale) >= c && numel(bias) >= c && numel(mean) >= c && numel(var) >= c,
                 "batch_norm: per-channel parameters shorter than C");
    LELE_TRY(ctx->arena_reset());
    const void *dx, *ds, *db, *dm, *dv;
    LELE_TRY(ctx->dev_ptr(x, &dx));
    LELE_TRY(ctx->dev_ptr(scale, &ds));
    LELE_TRY(ctx->dev_ptr(bias, &db));
    LELE_TRY(ctx->dev_ptr(mean, &dm));
    LELE_TRY(ctx->dev_ptr(var, &dv));
    LELE_TRY(out->reserve((size_t)n * 4));
    if (n) {
        hipLaunchKernelGGL(batch_norm_kernel, dim3(grid_for(n)), dim3(256), 0, ctx->stream, (const float*)dx,
                           (const float*)ds, (const float*)db, (const float*)dm, (const float*)dv, epsilon, c, inner, n,
                           (float*)out->data);
        LELE_HIP_CHECK(hipGetLastError());
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(x->shape, x->shape + x->rank));
}

}  // extern "C"
