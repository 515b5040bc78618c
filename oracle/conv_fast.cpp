// oracle/conv_fast.cpp -- lele's x86 Conv2d route as it RUNS: im2col + GEMM + a bias / activation pass
// (TEST INFRASTRUCTURE, see oracle.h: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
//
//   orc_conv2d_im2col  <- /root/reference/src/kernels/conv2d.rs:597-760: a thread-local column buffer [ICg*kh*kw][OH*OW] filled by
//                         im2col (conv2d.rs:892-1046: zero padding, row copies for stride 1, the gather loop otherwise), the
//                         group's weights [OCg][ICg*kh*kw] times that matrix (faer_matmul, Par::Seq), then per output channel
//                         bias + {none, ReLU, SiLU} in place (avx/math.rs bias_*_inplace: 8-wide polynomial body, libm tail).
//                         Depth-wise 3 x 3 / stride 1 / pad 1 layers take conv2d.rs:3131-3384's direct loop instead; here they run
//                         the plain loop of orc_conv2d (their share of a Yolo forward is 0.3 % of the multiply-adds).
//
// faer is not in the tree: the product here is the oracle's own AVX2-FMA kernel (fast.cpp: every element one
// k-ordered FMA chain), stated as such wherever this routine is timed.  Results agree with orc_conv2d (float64 accumulation) to
// f32 round-off; tests/test_oracle_fast.py pins that.
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <vector>

#include "oracle.h"

extern "C" void orc_fast_sgemm_kord(const float* a, int64_t lda, const float* b, int64_t ldb, int64_t m, int64_t k, int64_t n,
                                    float* c, int64_t ldc);

namespace {

// conv2d.rs:892-1046 (values only: which of its three loops fills a row does not change what is stored)
void im2col(const float* x, int64_t ih, int64_t iw, int64_t channels, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t pt,
            int64_t pl, int64_t dh, int64_t dw, int64_t oh, int64_t ow, float* col) {
    const int64_t plane = oh * ow;
    for (int64_t c = 0; c < channels; ++c)
        for (int64_t a = 0; a < kh; ++a)
            for (int64_t b = 0; b < kw; ++b) {
                float* row = col + ((c * kh + a) * kw + b) * plane;
                for (int64_t y = 0; y < oh; ++y) {
                    const int64_t iy = y * sh + a * dh - pt;
                    float* dst = row + y * ow;
                    if (iy < 0 || iy >= ih) {
                        memset(dst, 0, sizeof(float) * ow);
                        continue;
                    }
                    const float* src = x + (c * ih + iy) * iw;
                    if (sw == 1 && dw == 1) {
                        // columns xx with 0 <= xx + b - pl < iw are copies, the rest zeros
                        int64_t x0 = pl - b;
                        if (x0 < 0) x0 = 0;
                        int64_t x1 = iw + pl - b;
                        if (x1 > ow) x1 = ow;
                        if (x1 < x0) x1 = x0;
                        if (x0 > 0) memset(dst, 0, sizeof(float) * (x0 < ow ? x0 : ow));
                        if (x1 > x0) memcpy(dst + x0, src + x0 + b - pl, sizeof(float) * (x1 - x0));
                        if (x1 < ow) memset(dst + x1, 0, sizeof(float) * (ow - x1));
                    } else {
                        for (int64_t xx = 0; xx < ow; ++xx) {
                            const int64_t ix = xx * sw + b * dw - pl;
                            dst[xx] = ix >= 0 && ix < iw ? src[ix] : 0.0f;
                        }
                    }
                }
            }
}

}  // namespace

extern "C" void orc_conv2d_im2col(const float* x, const float* w, const float* bias, int64_t n, int64_t c, int64_t ih, int64_t iw,
                                  int64_t oc, int64_t kh, int64_t kw, int64_t group, int64_t pt, int64_t pl, int64_t pb, int64_t pr,
                                  int64_t sh, int64_t sw, int64_t dh, int64_t dw, int act, float* out) {
    const int64_t icg = c / group, ocg = oc / group;
    if (icg == 1 && ocg == 1) {  // depth-wise: no GEMM shape
        orc_conv2d(x, w, bias, n, c, ih, iw, oc, kh, kw, group, pt, pl, pb, pr, sh, sw, dh, dw, act, out);
        return;
    }
    const int64_t oh = (ih + pt + pb - dh * (kh - 1) - 1) / sh + 1, ow = (iw + pl + pr - dw * (kw - 1) - 1) / sw + 1;
    const int64_t plane = oh * ow, K = icg * kh * kw;
    static thread_local std::vector<float> col;  // conv2d.rs:601-603: COL_BUF
    if ((int64_t)col.size() < K * plane) col.resize((size_t)(K * plane));
    const bool pointwise = kh == 1 && kw == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0 && pb == 0 && pr == 0;
    for (int64_t b = 0; b < n; ++b)
        for (int64_t g = 0; g < group; ++g) {
            const float* xg = x + (b * c + g * icg) * ih * iw;
            const float* cm = xg;  // a 1 x 1 convolution's column matrix IS the input (conv2d.rs:297-420 skips the copy)
            if (!pointwise) {
                im2col(xg, ih, iw, icg, kh, kw, sh, sw, pt, pl, dh, dw, oh, ow, col.data());
                cm = col.data();
            }
            float* og = out + (b * oc + g * ocg) * plane;
            orc_fast_sgemm_kord(w + g * ocg * K, K, cm, plane, ocg, K, plane, og, plane);
            for (int64_t o = 0; o < ocg; ++o) {
                float* op = og + o * plane;
                const float bv = bias ? bias[g * ocg + o] : 0.0f;
                if (bv != 0.0f)  // conv2d.rs:693-733: the bias pass is skipped for a bias of exactly 0.0
                    for (int64_t i = 0; i < plane; ++i) op[i] = op[i] + bv;
                if (act == 1)
                    for (int64_t i = 0; i < plane; ++i) op[i] = op[i] > 0.0f ? op[i] : 0.0f;
                else if (act == 2)
                    orc_unary_simd(3, op, op, plane);
            }
        }
}
