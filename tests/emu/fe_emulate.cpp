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
    std::vector<cf> a(kLanes * kRegs);
    for (int p = 0; p < kLanes; ++p) {
        cf* ap = &a[p * kRegs];
        float x[kRegs];
        for (int r = 0; r < kRegs; ++r) x[r] = frame[16 * rev5(r) + p];
        if (use_fast) {
            phase_a_fast(x, ap, tw_re, tw_im);
        } else {
            for (int r = 0; r < kRegs; ++r) ap[r] = cmk(x[r], 0.0f);
            phase_a(ap, tw_re, tw_im);
        }
    }
    for (int k = 0; k < kBins; ++k) out_re[k] = out_im[k] = nanf("");
    auto twf = [&](int i) { return cmk(tw_re[i], tw_im[i]); };
    for (int rho = 0; rho < 2; ++rho) {
        // LDS image of one exchange round: 256 slots of (re, im)
        std::vector<cf> lds(16 * kXchgPitch, cmk(nanf(""), nanf("")));
        for (int p = 0; p < kLanes; ++p) {
            int h = rev4(p);
            for (int cc = 0; cc < 16; ++cc) lds[xchg_slot(h, cc)] = a[p * kRegs + rho * 16 + cc];
        }
        for (int c = 0; c < kLanes; ++c) {
            cf b[16];
            for (int v = 0; v < 16; ++v) b[v] = lds[xchg_slot(v, c)];
            float re256 = 0.0f;
            phase_b(b, c, rho, twf, &re256);
            for (int v = 0; v < 8; ++v) {
                int j = (2 * v + rho) * 16 + c;
                out_re[j] = b[v].x;
                out_im[j] = b[v].y;
            }
            if (c == 0 && rho == 0) {
                out_re[256] = re256;
                out_im[256] = 0.0f;
                out_im[0] = 0.0f;  // kernels/fft.rs:256-257
            }
        }
    }
}
