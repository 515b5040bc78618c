// gemm.hip -- lele::kernels::{matmul, matmul_fused_add, gemm} on the f32 MFMA core (gemm_core.h).
//
//   lele_hip_matmul            <- /root/reference/src/kernels/gemm.rs:112-222
//   lele_hip_matmul_fused_add  <- /root/reference/src/kernels/gemm.rs:223-432
//   lele_hip_gemm              <- /root/reference/src/kernels/gemm.rs:433-535
// Shapes, batching/broadcast rules, bias and beta*C broadcast cases and the panics (-> error codes) follow the
// reference; the inner product is the exact-f32 MFMA chain (faer's own summation order is not pinned).
#include "common.h"
#include "gemm_core.h"
#include "gemm_small.h"

using namespace lele;

namespace {

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

struct MmShape {
    int64_t batch_a, batch_b, fb, m, k, n;
    std::vector<int64_t> out_shape;
};

int mm_shape(const LeleTensor* a, const LeleTensor* b, MmShape* s, const char* who) {
    LELE_REQUIRE(a->rank >= 2 && b->rank >= 2, "%s: operands must have rank >= 2", who);  // gemm.rs:122-123
    LELE_REQUIRE(a->dtype == LELE_F32 && b->dtype == LELE_F32, "%s: operands must be f32", who);
    s->m = a->shape[a->rank - 2];
    s->k = a->shape[a->rank - 1];
    const int64_t kb = b->shape[b->rank - 2];
    s->n = b->shape[b->rank - 1];
    LELE_REQUIRE(s->k == kb, "MatMul K dim mismatch: %lld vs %lld", (long long)s->k, (long long)kb);  // gemm.rs:129
    s->batch_a = 1;
    for (int i = 0; i + 2 < a->rank; ++i) s->batch_a *= a->shape[i];
    s->batch_b = 1;
    for (int i = 0; i + 2 < b->rank; ++i) s->batch_b *= b->shape[i];
    s->fb = std::max(s->batch_a, s->batch_b);
    s->out_shape.clear();
    if (s->batch_a >= s->batch_b)
        s->out_shape.assign(a->shape, a->shape + a->rank - 2);
    else
        s->out_shape.assign(b->shape, b->shape + b->rank - 2);
    s->out_shape.push_back(s->m);
    s->out_shape.push_back(s->n);
    return 0;
}

// the small-problem kernel publishes one {min, max} pair per workgroup next to the result (LeleBuf::rowstat, kind 1)
float* stat_target(LeleCtx* ctx, LeleBuf* out, int64_t m, int64_t n, int64_t k, int64_t batch, int64_t* count) {
    *count = gemm::small_kernel_blocks((int)m, (int)n, (int)k, (int)batch, ctx->num_cus);
    if (*count <= 0 || *count > 4096 || lab_env("LELE_HIP_GEMM_FORCE")) return nullptr;
    if (m <= 4 || n <= 4) return nullptr;  // matrix-vector shapes go to gemm_f32_thin_kernel, which publishes nothing
    if (out->reserve_rowstat(*count) != 0 || (size_t)*count > out->rowstat_cap) return nullptr;
    return out->rowstat;
}
void stat_publish(LeleBuf* out, float* target, int64_t count, int64_t elements) {
    if (!target) return;
    out->rowstat_rows = count;
    out->rowstat_len = elements;
    out->rowstat_kind = 1;
    out->rowstat_valid = true;
}

int run_matmul(LeleCtx* ctx, const float* da, const float* db, const MmShape& s, float* out, float alpha, float beta,
               const float* c, int cmode, int64_t clen) {
    gemm::LoadRowK al{da, s.batch_a == 1 ? 0 : s.m * s.k, s.k, (int)s.m, (int)s.k,
                      (int)(aligned16(da) && s.k % 4 == 0)};
    gemm::LoadKRow bl{db, s.batch_b == 1 ? 0 : s.k * s.n, s.n, (int)s.n, (int)s.k};
    gemm::EpiAffine epi{out, s.m * s.n, (int)s.m, (int)s.n, alpha, beta, c, cmode, clen};
    gemm::launch(ctx->stream, al, bl, epi, (int)s.m, (int)s.n, (int)s.k, (int)s.fb, ctx->num_cus);
    LELE_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

extern "C" {

int lele_hip_matmul(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, LeleBuf* out, int64_t* out_shape,
                    int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && b && out, "matmul: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    MmShape s;
    LELE_TRY(mm_shape(a, b, &s, "matmul"));
    LELE_REQUIRE(s.batch_b == 1 || s.batch_b == s.batch_a, "MatMul broadcast not fully supported yet");  // gemm.rs:134
    LELE_TRY(ctx->arena_reset());
    const void *da = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    LELE_TRY(out->reserve((size_t)s.fb * s.m * s.n * 4));
    LELE_TRY(run_matmul(ctx, (const float*)da, (const float*)db, s, (float*)out->data, 1.0f, 0.0f, nullptr,
                        gemm::C_NONE, 0));
    return set_shape_v(out_shape, out_rank, s.out_shape);
}

int lele_hip_matmul_fused_add(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, const LeleTensor* bias,
                              LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && b && bias && out, "matmul_fused_add: NULL argument");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    MmShape s;
    LELE_TRY(mm_shape(a, b, &s, "matmul_fused_add"));
    LELE_REQUIRE(s.batch_b == 1 || s.batch_b == s.batch_a || s.batch_a == 1,
                 "matmul_fused_add: batch dims %lld vs %lld cannot be broadcast", (long long)s.batch_a,
                 (long long)s.batch_b);
    const int64_t blen = numel(bias);
    LELE_REQUIRE(blen > 0 && bias->dtype == LELE_F32, "matmul_fused_add: bias must be a non-empty f32 tensor");
    LELE_TRY(ctx->arena_reset());
    const void *da = nullptr, *db = nullptr, *dc = nullptr;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    LELE_TRY(ctx->dev_ptr(bias, &dc));
    LELE_TRY(out->reserve((size_t)s.fb * s.m * s.n * 4));
    // bias.len() == n: rows prefilled with the bias (gemm.rs:251-316); otherwise out[i] += bias[i % len] over the
    // flattened result (gemm.rs:354-415)
    const int cmode = (blen == s.n) ? gemm::C_ROWVEC : gemm::C_MODULO;
    LELE_TRY(run_matmul(ctx, (const float*)da, (const float*)db, s, (float*)out->data, 1.0f, 1.0f, (const float*)dc,
                        cmode, blen));
    return set_shape_v(out_shape, out_rank, s.out_shape);
}

int lele_hip_gemm(LeleCtx* ctx, const LeleTensor* a, const LeleTensor* b, const LeleTensor* c, float alpha, float beta,
                  int trans_a, int trans_b, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && b && out, "gemm: NULL argument");
    LELE_REQUIRE(a->rank >= 2 && b->rank >= 2, "gemm: operands must have rank >= 2");
    LELE_REQUIRE(a->dtype == LELE_F32 && b->dtype == LELE_F32, "gemm: operands must be f32");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    const int64_t m = trans_a ? a->shape[a->rank - 1] : a->shape[a->rank - 2];  // gemm.rs:446-465
    const int64_t k = trans_a ? a->shape[a->rank - 2] : a->shape[a->rank - 1];
    const int64_t n = trans_b ? b->shape[b->rank - 2] : b->shape[b->rank - 1];
    const int64_t k2 = trans_b ? b->shape[b->rank - 1] : b->shape[b->rank - 2];
    LELE_REQUIRE(k == k2, "Gemm K dim mismatch");  // gemm.rs:466
    LELE_TRY(ctx->arena_reset());
    const void *da = nullptr, *db = nullptr, *dc = nullptr;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    int cmode = gemm::C_NONE;
    int64_t clen = 0;
    if (c && beta != 0.0f) {  // gemm.rs:484-515
        clen = numel(c);
        LELE_REQUIRE(clen > 0, "gemm: empty C");
        LELE_TRY(ctx->dev_ptr(c, &dc));
        if (clen == m * n)
            cmode = gemm::C_FULL;
        else if (clen == n)
            cmode = gemm::C_ROWVEC;
        else if (clen == m)
            cmode = gemm::C_COLVEC;
        else if (clen == 1)
            cmode = gemm::C_SCALAR;
        else
            cmode = gemm::C_MODULO;
    }
    LELE_TRY(out->reserve((size_t)m * n * 4));
    const float* fa = (const float*)da;
    const float* fb = (const float*)db;
    gemm::EpiAffine epi{(float*)out->data, m * n, (int)m, (int)n, alpha, beta, (const float*)dc, cmode, clen};
    // transposes are strides (gemm.rs:517-520): A(m,k) = a[m*rsa + k*csa], B(k,n) = b[k*rsb + n*csb]
    gemm::LoadRowK a_mk{fa, 0, k, (int)m, (int)k, (int)(aligned16(fa) && k % 4 == 0)};
    gemm::LoadKRow a_km{fa, 0, m, (int)m, (int)k};
    gemm::LoadKRow b_kn{fb, 0, n, (int)n, (int)k};
    gemm::LoadRowK b_nk{fb, 0, k, (int)n, (int)k, (int)(aligned16(fb) && k % 4 == 0)};
    if (!trans_a && !trans_b)
        gemm::launch(ctx->stream, a_mk, b_kn, epi, (int)m, (int)n, (int)k, 1, ctx->num_cus);
    else if (!trans_a && trans_b)
        gemm::launch(ctx->stream, a_mk, b_nk, epi, (int)m, (int)n, (int)k, 1, ctx->num_cus);
    else if (trans_a && !trans_b)
        gemm::launch(ctx->stream, a_km, b_kn, epi, (int)m, (int)n, (int)k, 1, ctx->num_cus);
    else
        gemm::launch(ctx->stream, a_km, b_nk, epi, (int)m, (int)n, (int)k, 1, ctx->num_cus);
    LELE_HIP_CHECK(hipGetLastError());
    return set_shape(out_shape, out_rank, {m, n});
}

int lele_hip_matmul_view(LeleCtx* ctx, const LeleTensor* a, const LeleMatView* av, const LeleTensor* b, const LeleMatView* bv,
                         int64_t batch_outer, int64_t batch_inner, int64_t m, int64_t k, int64_t n, const LeleMatView* ov,
                         const int64_t* out_dims, int32_t out_dims_rank, LeleBuf* out, int64_t* out_shape, int32_t* out_rank) {
    LELE_REQUIRE(ctx && a && av && b && bv && ov && out && out_dims, "matmul_view: NULL argument");
    LELE_REQUIRE(a->dtype == LELE_F32 && b->dtype == LELE_F32, "matmul_view: operands must be f32");
    LELE_REQUIRE(batch_outer >= 1 && batch_inner >= 1 && m >= 0 && n >= 0 && k >= 0, "matmul_view: bad dimensions");
    LELE_REQUIRE(av->stride_col == 1 || av->stride_row == 1, "matmul_view: A must be contiguous along k or along its rows");
    LELE_REQUIRE(bv->stride_col == 1 || bv->stride_row == 1, "matmul_view: B must be contiguous along n or along k");
    LELE_REQUIRE(ov->stride_col == 1, "matmul_view: the output must be contiguous along n");
    const int64_t fb = batch_outer * batch_inner, na = numel(a), nb = numel(b);
    int64_t total = 1;
    for (int i = 0; i < out_dims_rank; ++i) total *= out_dims[i];
    LELE_REQUIRE(total == fb * m * n, "matmul_view: output shape holds %lld elements, the product has %lld", (long long)total,
                 (long long)(fb * m * n));
    LELE_REQUIRE(fb < (int64_t(1) << 31) && m < (int64_t(1) << 31) && n < (int64_t(1) << 31) && k < (int64_t(1) << 31), "matmul_view: dimension too large");
    // the last element each view can touch must lie inside its tensor
    auto last = [&](const LeleMatView* v, int64_t r, int64_t c) {
        return v->offset + (batch_outer - 1) * v->stride_outer + (batch_inner - 1) * v->stride_inner + (r - 1) * v->stride_row + (c - 1) * v->stride_col;
    };
    if (fb * m * k) LELE_REQUIRE(av->offset >= 0 && last(av, m, k) < na, "matmul_view: the A view leaves its tensor");
    if (fb * k * n) LELE_REQUIRE(bv->offset >= 0 && last(bv, k, n) < nb, "matmul_view: the B view leaves its tensor");
    if (total) LELE_REQUIRE(ov->offset >= 0 && last(ov, m, n) < total, "matmul_view: the output view leaves its buffer");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_TRY(ctx->arena_reset());
    const void *da = nullptr, *db = nullptr;
    LELE_TRY(ctx->dev_ptr(a, &da));
    LELE_TRY(ctx->dev_ptr(b, &db));
    LELE_TRY(out->reserve((size_t)total * 4));
    if (total) {
        const float* pa = (const float*)da + av->offset;
        const float* pb = (const float*)db + bv->offset;
        const int bi = (int)batch_inner;
        gemm::EpiAffine epi{(float*)out->data + ov->offset, ov->stride_outer, (int)m, (int)n, 1.0f, 0.0f, nullptr, gemm::C_NONE, 0,
                            ov->stride_row, bi, ov->stride_inner};
        int64_t nstat = 0;
        epi.blockstat = stat_target(ctx, out, m, n, k, fb, &nstat);
        // A(row, k): k-contiguous -> LoadRowK(ld = row stride); row-contiguous -> LoadKRow(ld = k stride).  B likewise on (n, k).
        gemm::LoadRowK a_rk{pa, av->stride_outer, av->stride_row, (int)m, (int)k,
                            (int)(aligned16(pa) && av->stride_row % 4 == 0 && av->stride_outer % 4 == 0 && av->stride_inner % 4 == 0), bi, av->stride_inner};
        gemm::LoadKRow a_kr{pa, av->stride_outer, av->stride_col, (int)m, (int)k, bi, av->stride_inner};
        gemm::LoadKRow b_kn{pb, bv->stride_outer, bv->stride_row, (int)n, (int)k, bi, bv->stride_inner};
        gemm::LoadRowK b_nk{pb, bv->stride_outer, bv->stride_col, (int)n, (int)k,
                            (int)(aligned16(pb) && bv->stride_col % 4 == 0 && bv->stride_outer % 4 == 0 && bv->stride_inner % 4 == 0), bi, bv->stride_inner};
        const bool a_k = av->stride_col == 1, b_n = bv->stride_col == 1;
        if (a_k && b_n) gemm::launch(ctx->stream, a_rk, b_kn, epi, (int)m, (int)n, (int)k, (int)fb, ctx->num_cus);
        else if (a_k && !b_n) gemm::launch(ctx->stream, a_rk, b_nk, epi, (int)m, (int)n, (int)k, (int)fb, ctx->num_cus);
        else if (!a_k && b_n) gemm::launch(ctx->stream, a_kr, b_kn, epi, (int)m, (int)n, (int)k, (int)fb, ctx->num_cus);
        else gemm::launch(ctx->stream, a_kr, b_nk, epi, (int)m, (int)n, (int)k, (int)fb, ctx->num_cus);
        LELE_HIP_CHECK(hipGetLastError());
        stat_publish(out, epi.blockstat, nstat, total);
    }
    return set_shape_v(out_shape, out_rank, std::vector<int64_t>(out_dims, out_dims + out_dims_rank));
}

}  // extern "C"