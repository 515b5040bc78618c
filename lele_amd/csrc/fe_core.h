// fe_core.h -- per-lane arithmetic of the 512-point front-end FFT, shared by the HIP kernel
// (frontend.hip) and by the host-side lane emulator used in tests (tests/emu/fe_emulate.cpp).
//
// The reference computes a full 512-point complex radix-2 DIT FFT on the real frame
// (/root/reference/src/kernels/fft.rs:172-266, x86 AVX2 branch).  Its f32 round-off noise is
// comparable to the weakest bins of a high-dynamic-range frame, so agreeing with it to 1e-4
// requires reproducing the SAME butterfly network with the SAME roundings:
//   stages 1,2 (half_size 1,2  -> scalar tail, fft.rs:236-250):  tr = wr*or - wi*oi (two roundings + sub)
//   stages 3..9 (half_size>=4 -> SSE/AVX, fft.rs:192-234):        tr = fma(wr,or, -(wi*oi)), ti = fma(wr,oi, wi*or)
// and the same twiddle table values (cosf/sinf of the f32 angle, fft.rs:136-157).
//
// Mapping of the 512 positions j (after bit reversal) onto a 16-lane group holding 32 points per lane:
//   phase A  lane p (0..15), h = rev4(p) = j[8:5], register r = j[4:0]; input sample n = 16*rev5(r) + p.
//            stages 1..5 pair register bits 0..4; twiddle index k = r mod half (wave-uniform).
//   exchange (LDS), two rounds rho = r>>4:  register r of lane h  ->  lane c = r&15, register v = h.
//   phase B  lane c = j[3:0], register v = j[8:5], rho = j[4]; stages 6..9 pair register bits 0..3 of v;
//            twiddle index k = (v mod 2^(s-6))*32 + rho*16 + c (per lane, fetched from the table).
//            Only bins 0..256 are needed: stage 9 computes the '+' outputs (v<8) and, for (c=0,rho=0), bin 256.
//
// Complex values are (re, im) pairs in ONE two-element vector so that gfx950's packed-f32 VALU ops
// (v_pk_mul/add/fma_f32, each IEEE-exact per element) do a whole complex add or half a complex multiply per
// instruction; the host build uses a plain struct with the same element-wise semantics.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FE_HD __host__ __device__ __forceinline__
#else
#define FE_HD inline
#endif

namespace fe {

constexpr int kNfft = 512;
constexpr int kFrame = 400;
constexpr int kHop = 160;
constexpr int kBins = 257;
constexpr int kLanes = 16;   // lanes per frame
constexpr int kRegs = 32;    // points per lane
constexpr int kQ = 25;       // samples per lane that are inside the 400-sample frame (q = 0..24)

FE_HD constexpr int rev4(int x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }
FE_HD constexpr int rev5(int x) {
    return ((x & 1) << 4) | ((x & 2) << 2) | (x & 4) | ((x & 8) >> 2) | ((x & 16) >> 4);
}
// offset of stage s (1-based) inside the concatenated twiddle table of precompute_twiddles()
FE_HD constexpr int tw_off(int s) { return (1 << (s - 1)) - 1; }

// explicit roundings: the translation unit is built with -ffp-contract=off, fma only where written.
FE_HD float fmul(float a, float b) { return a * b; }
FE_HD float fadd(float a, float b) { return a + b; }
FE_HD float fsub(float a, float b) { return a - b; }
FE_HD float ffma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
typedef float cf __attribute__((ext_vector_type(2)));
FE_HD cf cmk(float re, float im) { return (cf){re, im}; }
FE_HD cf cadd(cf a, cf b) { return a + b; }
FE_HD cf csub(cf a, cf b) { return a - b; }
FE_HD cf cmulv(cf a, cf b) { return a * b; }  // element-wise
FE_HD cf cfmav(cf a, cf b, cf c) { return __builtin_elementwise_fma(a, b, c); }
FE_HD cf cswap(cf a) { return a.yx; }
FE_HD cf cxx(cf a) { return a.xx; }
#else
struct cf {
    float x, y;
};
FE_HD cf cmk(float re, float im) { return cf{re, im}; }
FE_HD cf cadd(cf a, cf b) { return cf{a.x + b.x, a.y + b.y}; }
FE_HD cf csub(cf a, cf b) { return cf{a.x - b.x, a.y - b.y}; }
FE_HD cf cmulv(cf a, cf b) { return cf{a.x * b.x, a.y * b.y}; }
FE_HD cf cfmav(cf a, cf b, cf c) { return cf{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)}; }
FE_HD cf cswap(cf a) { return cf{a.y, a.x}; }
FE_HD cf cxx(cf a) { return cf{a.x, a.x}; }
#endif

// Twiddle in the two forms the butterfly consumes: w = (wr, wi) and wn = (-wi, wi).
struct Tw {
    cf w, wn;
};
FE_HD Tw make_tw(float wr, float wi) { return Tw{cmk(wr, wi), cmk(-wi, wi)}; }

// t = w * o with the reference's SIMD roundings (fft.rs:200-201):
//   t.re = fma(wr, o.re, -(wi*o.im))   t.im = fma(wr, o.im, wi*o.re)
FE_HD cf tw_mul_fma(const Tw& t, cf o) { return cfmav(cxx(t.w), o, cmulv(t.wn, cswap(o))); }

// Generic butterfly of the SIMD stages (fft.rs:200-208)
FE_HD void bfly_fma(const Tw& tw, cf& e, cf& o) {
    cf t = tw_mul_fma(tw, o);
    cf n = csub(e, t);
    e = cadd(e, t);
    o = n;
}
// Scalar-tail butterfly (fft.rs:236-250): both products rounded, then add/sub.
FE_HD void bfly_scalar(float wr, float wi, cf& e, cf& o) {
    float tr = fsub(fmul(wr, o.x), fmul(wi, o.y));
    float ti = fadd(fmul(wr, o.y), fmul(wi, o.x));
    cf t = cmk(tr, ti);
    cf n = csub(e, t);
    e = cadd(e, t);
    o = n;
}
// scalar-argument forms used by the generic any-n FFT kernel (features_ops.hip)
FE_HD void bfly_fma(float wr, float wi, float& er, float& ei, float& o_r, float& oi) {
    cf e = cmk(er, ei), o = cmk(o_r, oi);
    bfly_fma(make_tw(wr, wi), e, o);
    er = e.x;
    ei = e.y;
    o_r = o.x;
    oi = o.y;
}
FE_HD void bfly_scalar(float wr, float wi, float& er, float& ei, float& o_r, float& oi) {
    cf e = cmk(er, ei), o = cmk(o_r, oi);
    bfly_scalar(wr, wi, e, o);
    er = e.x;
    ei = e.y;
    o_r = o.x;
    oi = o.y;
}

// ---- phase A: stages 1..5 on the 32 register-resident points of one lane ---------------------------
// a[r] (r = 0..31).  On entry a[r] = (windowed sample n = 16*rev5(r)+p, 0); samples with rev5(r) >= 25 are 0.
// tw_re/tw_im: the reference's concatenated twiddle table (511 entries).
template <typename TW>
FE_HD void phase_a(cf* a, const TW& tw_re, const TW& tw_im) {
    // stage 1 (half 1, scalar path, w = table[0] = (1, -0))
#pragma unroll
    for (int r = 0; r < kRegs; r += 2) bfly_scalar(tw_re[tw_off(1)], tw_im[tw_off(1)], a[r], a[r + 1]);
    // stage 2 (half 2, scalar path)
#pragma unroll
    for (int b = 0; b < kRegs; b += 4) {
#pragma unroll
        for (int k = 0; k < 2; ++k) bfly_scalar(tw_re[tw_off(2) + k], tw_im[tw_off(2) + k], a[b + k], a[b + 2 + k]);
    }
    // stages 3..5 (half 4,8,16 -> SSE/AVX path with fma)
#pragma unroll
    for (int s = 3; s <= 5; ++s) {
        const int half = 1 << (s - 1);
#pragma unroll
        for (int b = 0; b < kRegs; b += 2 * half) {
#pragma unroll
            for (int k = 0; k < half; ++k)
                bfly_fma(make_tw(tw_re[tw_off(s) + k], tw_im[tw_off(s) + k]), a[b + k], a[b + half + k]);
        }
    }
}

// Same network with the structural zeros folded away (values identical to phase_a up to the sign of zeros):
//   * stage 1 and stage 2/k=0 use table[0] = table[1] = (1, -0): x*1 == x, x - (+-0) == x, so they are
//     real add/sub while the imaginary parts are still exactly zero;
//   * registers whose sample index 16*rev5(r)+p is >= 400 are the zero padding: stage-1 butterflies whose
//     odd input is padding reduce to copies (e + 0, e - 0);
//   * stage 2/k=1 and stage 3/k even have zero imaginary inputs: fma(w, 0, x) == x, w*0 == +-0.
// REQUIRES tw_re[0]==1, tw_im[0]==+-0, tw_re[1]==1, tw_im[1]==+-0 (checked on the host when tables are built).
// Input: x[r] real (r = 0..31, zero where rev5(r) >= 25).  Output: a[r] complex.
template <typename TW>
FE_HD void phase_a_fast(const float* x, cf* a, const TW& tw_re, const TW& tw_im) {
    float s1[kRegs];
    // stage 1
#pragma unroll
    for (int r = 0; r < kRegs; r += 2) {
        if (rev5(r + 1) >= kQ) {  // odd input is zero padding (compile-time condition)
            s1[r] = x[r];
            s1[r + 1] = x[r];
        } else {
            s1[r] = fadd(x[r], x[r + 1]);
            s1[r + 1] = fsub(x[r], x[r + 1]);
        }
    }
    // stage 2: k=0 real add/sub; k=1 twiddle table[2] on a real odd input
    {
        const float wr = tw_re[tw_off(2) + 1], wi = tw_im[tw_off(2) + 1];
#pragma unroll
        for (int b = 0; b < kRegs; b += 4) {
            a[b] = cmk(fadd(s1[b], s1[b + 2]), 0.0f);
            a[b + 2] = cmk(fsub(s1[b], s1[b + 2]), 0.0f);
            const float tr = fmul(wr, s1[b + 3]);  // wr*or - wi*0
            const float ti = fmul(wi, s1[b + 3]);  // wr*0 + wi*or
            a[b + 1] = cmk(fadd(s1[b + 1], tr), ti);   // (e + tr, 0 + ti)
            a[b + 3] = cmk(fsub(s1[b + 1], tr), -ti);  // (e - tr, 0 - ti)
        }
    }
    // stage 3: k = 0,2 have purely real inputs
#pragma unroll
    for (int b = 0; b < kRegs; b += 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wr = tw_re[tw_off(3) + k], wi = tw_im[tw_off(3) + k];
            if ((k & 1) == 0) {
                const float e = a[b + k].x, o = a[b + 4 + k].x;
                const float tr = fmul(wr, o);  // fma(wr, or, -(wi*0))
                const float ti = fmul(wi, o);  // fma(wr, 0, wi*or)
                a[b + k] = cmk(fadd(e, tr), ti);
                a[b + 4 + k] = cmk(fsub(e, tr), -ti);
            } else {
                bfly_fma(make_tw(wr, wi), a[b + k], a[b + 4 + k]);
            }
        }
    }
#pragma unroll
    for (int s = 4; s <= 5; ++s) {
        const int half = 1 << (s - 1);
#pragma unroll
        for (int b = 0; b < kRegs; b += 2 * half) {
#pragma unroll
            for (int k = 0; k < half; ++k)
                bfly_fma(make_tw(tw_re[tw_off(s) + k], tw_im[tw_off(s) + k]), a[b + k], a[b + half + k]);
        }
    }
}

// ---- phase B: stages 6..9 on 16 points (register v = j[8:5]) of lane c, round rho -------------------
// Twiddles are per lane: index = tw_off(s) + (v mod 2^(s-6))*32 + rho*16 + c; TWF(i) returns (wr, wi) of entry i.
// After the call: b[v] for v<8 hold bins j = (2v+rho)*16 + c; *re256 = re of position 256 (meaningful for
// c==0, rho==0 only).
template <typename TWF>
FE_HD void phase_b(cf* b, int c, int rho, const TWF& twf, float* re256) {
#pragma unroll
    for (int s = 6; s <= 8; ++s) {
        const int hv = 1 << (s - 6);  // half size in units of v
#pragma unroll
        for (int k = 0; k < hv; ++k) {
            const cf w = twf(tw_off(s) + k * 32 + rho * 16 + c);
            const Tw tw = make_tw(w.x, w.y);
#pragma unroll
            for (int base = 0; base < 16; base += 2 * hv) bfly_fma(tw, b[base + k], b[base + hv + k]);
        }
    }
    // stage 9: pairs v and v+8.  Bins <256 are the even ('+') outputs.
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const cf w = twf(tw_off(9) + k * 32 + rho * 16 + c);
        const cf t = tw_mul_fma(make_tw(w.x, w.y), b[8 + k]);
        if (k == 0) *re256 = fsub(b[0].x, t.x);  // position 256 = odd output of the k=0 butterfly
        b[k] = cadd(b[k], t);
    }
}

// power spectrum exactly as features/pipeline.rs:165-169 after kernels/fft.rs:256-265 forced
// im[0] = im[256] = 0:  re*re + im*im  (two products, one add, no fma).
FE_HD float power(float re, float im) { return fadd(fmul(re, re), fmul(im, im)); }
FE_HD float power(cf z) {
    cf sq = cmulv(z, z);
    return fadd(sq.x, sq.y);
}

// LDS slot (in 8-byte units) of element (v = writer h, c) of one round of the exchange, per frame.
// Row pitch 17 slots (136 B = 34 banks): the 16 writers (lanes h, fixed c) land on banks 2h (+1), the 16
// readers (lanes c, fixed v) read consecutive slots -- both conflict-free, and every address is
// lane_base + compile-time immediate (no per-access address arithmetic).
constexpr int kXchgPitch = 17;
FE_HD constexpr int xchg_slot(int v, int c) { return v * kXchgPitch + c; }

}  // namespace fe
