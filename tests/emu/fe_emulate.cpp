// tests/emu/fe_emulate.cpp -- host-side lane emulator for lele_amd/csrc/fe_core.h (test infrastructure).
// Runs the SAME per-lane code the HIP kernel runs (phase_a / exchange addressing / phase_b) for the 16
// lanes of one frame, sequentially, and returns the 257 complex bins.  tests/test_fe_core_emulation.py
// checks the result bit-for-bit against the oracle's restatement of the reference FFT.
#include <math.h>
#include <string.h>
#include <vector>
#include "../../lele_amd/csrc/fe_core.h"

extern "C" void fe_emulate_fft512(const float* frame /*512, zero padded*/, const float* tw_re, const float* tw_im,
                                  float* out_re /*257*/, float* out_im /*257*/, int use_fast) {
    using namespace fe;
    // LDS image of the exchange, one round at a time: 256 slots of (re, im)
    std::vector<float> are(kLanes * kRegs), aim(kLanes * kRegs, 0.0f);
    for (int p = 0; p < kLanes; ++p) {
        float* r_re = &are[p * kRegs];
        float* r_im = &aim[p * kRegs];
        for (int r = 0; r < kRegs; ++r) r_re[r] = frame[16 * rev5(r) + p];
        if (use_fast)
            phase_a_fast(r_re, r_im, tw_re, tw_im);
        else
            phase_a(r_re, r_im, tw_re, tw_im);
    }
    for (int k = 0; k < kBins; ++k) out_re[k] = out_im[k] = nanf("");
    for (int rho = 0; rho < 2; ++rho) {
        std::vector<Cplx> lds(256, Cplx{nanf(""), nanf("")});
        for (int p = 0; p < kLanes; ++p) {
            int h = rev4(p);
            for (int cc = 0; cc < 16; ++cc) {
                int r = rho * 16 + cc;
                lds[xchg_slot(h, cc)] = Cplx{are[p * kRegs + r], aim[p * kRegs + r]};
            }
        }
        for (int c = 0; c < kLanes; ++c) {
            float bre[16], bim[16];
            for (int v = 0; v < 16; ++v) {
                bre[v] = lds[xchg_slot(v, c)].re;
                bim[v] = lds[xchg_slot(v, c)].im;
            }
            phase_b(bre, bim, c, rho, tw_re, tw_im);
            for (int v = 0; v < 8; ++v) {
                int j = (2 * v + rho) * 16 + c;
                out_re[j] = bre[v];
                out_im[j] = bim[v];
            }
            if (c == 0 && rho == 0) {
                out_re[256] = bre[8];
                out_im[256] = 0.0f;
                out_im[0] = 0.0f;  // kernels/fft.rs:256-257
            }
        }
    }
}
