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

struct Cplx {
    float re, im;
};

// Generic butterfly of the SIMD stages (fft.rs:200-208): t = w*o with the reference's fma pattern.
FE_HD void bfly_fma(float wr, float wi, float& er, float& ei, float& o_r, float& oi) {
    float tr = ffma(wr, o_r, -fmul(wi, oi));  // _mm256_fmsub_ps(wr, od_re, _mm256_mul_ps(wi, od_im))
    float ti = ffma(wr, oi, fmul(wi, o_r));   // _mm256_fmadd_ps(wr, od_im, _mm256_mul_ps(wi, od_re))
    float nr = fsub(er, tr), ni = fsub(ei, ti);
    er = fadd(er, tr);
    ei = fadd(ei, ti);
    o_r = nr;
    oi = ni;
}
// Scalar-tail butterfly (fft.rs:236-250): both products rounded, then add/sub.
FE_HD void bfly_scalar(float wr, float wi, float& er, float& ei, float& o_r, float& oi) {
    float tr = fsub(fmul(wr, o_r), fmul(wi, oi));
    float ti = fadd(fmul(wr, oi), fmul(wi, o_r));
    float nr = fsub(er, tr), ni = fsub(ei, ti);
    er = fadd(er, tr);
    ei = fadd(ei, ti);
    o_r = nr;
    oi = ni;
}

// ---- phase A: stages 1..5 on the 32 register-resident points of one lane ---------------------------
// are/aim: [32], index = r.  On entry are[r] = windowed sample n = 16*rev5(r)+p (0 when rev5(r) >= 25),
// aim[r] = 0.  tw_re/tw_im: the reference's concatenated twiddle table (511 entries).
template <typename TW>
FE_HD void phase_a(float* are, float* aim, const TW& tw_re, const TW& tw_im) {
    // stage 1 (half 1, scalar path, w = table[0] = (1, -0)): with im == 0 this is a real add/sub.
    // The generic scalar formula gives exactly the same values (x*1, x - (+-0)), up to the sign of zeros.
#pragma unroll
    for (int r = 0; r < kRegs; r += 2) {
        bfly_scalar(tw_re[tw_off(1)], tw_im[tw_off(1)], are[r], aim[r], are[r + 1], aim[r + 1]);
    }
    // stage 2 (half 2, scalar path)
#pragma unroll
    for (int b = 0; b < kRegs; b += 4) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            bfly_scalar(tw_re[tw_off(2) + k], tw_im[tw_off(2) + k], are[b + k], aim[b + k], are[b + 2 + k],
                        aim[b + 2 + k]);
    }
    // stages 3..5 (half 4,8,16 -> SSE/AVX path with fma)
#pragma unroll
    for (int s = 3; s <= 5; ++s) {
        const int half = 1 << (s - 1);
#pragma unroll
        for (int b = 0; b < kRegs; b += 2 * half) {
#pragma unroll
            for (int k = 0; k < half; ++k)
                bfly_fma(tw_re[tw_off(s) + k], tw_im[tw_off(s) + k], are[b + k], aim[b + k], are[b + half + k],
                         aim[b + half + k]);
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
template <typename TW>
FE_HD void phase_a_fast(float* are, float* aim, const TW& tw_re, const TW& tw_im) {
    // stage 1
#pragma unroll
    for (int r = 0; r < kRegs; r += 2) {
        if (rev5(r + 1) >= kQ) {  // odd input is zero padding (compile-time condition)
            are[r + 1] = are[r];
        } else {
            float e = are[r], o = are[r + 1];
            are[r] = fadd(e, o);
            are[r + 1] = fsub(e, o);
        }
    }
    // stage 2: k=0 real add/sub; k=1 twiddle table[2] on a real odd input
    {
        const float wr = tw_re[tw_off(2) + 1], wi = tw_im[tw_off(2) + 1];
#pragma unroll
        for (int b = 0; b < kRegs; b += 4) {
            float e = are[b], o = are[b + 2];
            are[b] = fadd(e, o);
            are[b + 2] = fsub(e, o);
            float e1 = are[b + 1], o1 = are[b + 3];
            float tr = fmul(wr, o1);  // wr*or - wi*0
            float ti = fmul(wi, o1);  // wr*0 + wi*or
            are[b + 1] = fadd(e1, tr);
            are[b + 3] = fsub(e1, tr);
            aim[b + 1] = ti;   // 0 + ti
            aim[b + 3] = -ti;  // 0 - ti
        }
    }
    // stage 3: k = 0,2 have purely real inputs
#pragma unroll
    for (int b = 0; b < kRegs; b += 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float wr = tw_re[tw_off(3) + k], wi = tw_im[tw_off(3) + k];
            if ((k & 1) == 0) {
                float e = are[b + k], o = are[b + 4 + k];
                float tr = fmul(wr, o);  // fma(wr, or, -(wi*0))
                float ti = fmul(wi, o);  // fma(wr, 0, wi*or)
                are[b + k] = fadd(e, tr);
                are[b + 4 + k] = fsub(e, tr);
                aim[b + k] = ti;
                aim[b + 4 + k] = -ti;
            } else {
                bfly_fma(wr, wi, are[b + k], aim[b + k], are[b + 4 + k], aim[b + 4 + k]);
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
                bfly_fma(tw_re[tw_off(s) + k], tw_im[tw_off(s) + k], are[b + k], aim[b + k], are[b + half + k],
                         aim[b + half + k]);
        }
    }
}

// ---- phase B: stages 6..9 on 16 points (register v = j[8:5]) of lane c, round rho -------------------
// Twiddles are per lane: index = tw_off(s) + (v mod 2^(s-6))*32 + rho*16 + c.
// After the call: bre/bim[v] for v<8 hold bins j = (2v+rho)*16 + c; bre[8] (only meaningful for c==0,rho==0)
// holds re of position 256 (bin 256).
template <typename TW>
FE_HD void phase_b(float* bre, float* bim, int c, int rho, const TW& tw_re, const TW& tw_im) {
#pragma unroll
    for (int s = 6; s <= 8; ++s) {
        const int hv = 1 << (s - 6);  // half size in units of v
#pragma unroll
        for (int b = 0; b < 16; b += 2 * hv) {
#pragma unroll
            for (int k = 0; k < hv; ++k) {
                const int ti = tw_off(s) + k * 32 + rho * 16 + c;
                bfly_fma(tw_re[ti], tw_im[ti], bre[b + k], bim[b + k], bre[b + hv + k], bim[b + hv + k]);
            }
        }
    }
    // stage 9: pairs v and v+8.  Bins <256 are the even ('+') outputs.
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int ti = tw_off(9) + k * 32 + rho * 16 + c;
        const float wr = tw_re[ti], wi = tw_im[ti];
        float tr = ffma(wr, bre[8 + k], -fmul(wi, bim[8 + k]));
        float tim = ffma(wr, bim[8 + k], fmul(wi, bre[8 + k]));
        if (k == 0) {  // position 256 = odd output of the k=0 butterfly (only used when c==0 && rho==0)
            bre[8] = fsub(bre[0], tr);
        }
        bre[k] = fadd(bre[k], tr);
        bim[k] = fadd(bim[k], tim);
    }
}

// power spectrum exactly as features/pipeline.rs:165-169 after kernels/fft.rs:256-265 forced
// im[0] = im[256] = 0:  re*re + im*im  (two products, one add, no fma).
FE_HD float power(float re, float im) { return fadd(fmul(re, re), fmul(im, im)); }

// LDS slot (in 8-byte units) of element (v = writer h, c) of one round of the exchange, per frame.
// XOR keeps both the writers (16 lanes h, fixed c) and the readers (16 lanes c, fixed v) conflict-free.
FE_HD constexpr int xchg_slot(int v, int c) { return v * 16 + (c ^ v); }

}  // namespace fe
