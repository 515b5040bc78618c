#!/usr/bin/env python3
"""Phase timing of igemm_wholek_kernel from inside the kernel: thread 0 of every workgroup stamps the shader clock at the phase
boundaries of each chunk (developer switch LELE_HIP_WHOLEK_STAMPS = device address of a [grid][64] i64 buffer).  Prints, per
call shape, the median over workgroups of each phase in microseconds at the kernel's own clock (cycle counter, 100 MHz on gfx9)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from lele_amd import kernels as K
    from lele_amd._lib import Ctx, Weight
    ctx = Ctx()
    rng = np.random.default_rng(0)
    dbg = torch.zeros((256, 64), dtype=torch.int64, device="cuda")
    names = ["start", "first-data"] + [p + str(j) for j in range(6) for p in ("barrier", "mfma+stage", "epilogue")]
    abl = os.environ.get("LELE_HIP_WHOLEK_ABLATE", "0")
    print("ablate =", abl)
    for name, n, em in (("qkv (EM 0)", 1536, 0), ("ffn hidden, range pass (EM 1)", 2048, 1), ("ffn hidden, quantise pass (EM 2)", 2048, 2)):
        b, m, k = 32, 171, 512
        x = ctx.buf().upload(rng.standard_normal((b, m, k)).astype(np.float32))
        g, be = Weight(np.ones(k, np.float32)), Weight(np.zeros(k, np.float32))
        xn = K.layer_norm(x, g, be, -1, 1e-5, out=ctx.buf(), ctx=ctx)

        def lin(kk, nn):
            return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((kk, nn))), 0, 255).astype(np.float32)),
                    Weight((np.abs(rng.standard_normal(nn)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
                    Weight((rng.standard_normal(nn) * 0.02).astype(np.float32)))
        w, w2 = lin(k, n), lin(n, 512)
        r1 = ctx.buf().upload(rng.standard_normal((b, m, n)).astype(np.float32))
        ob = ctx.buf()
        if em:
            call = lambda: K.fused_ffn_quantized(xn, *w, *w2, False, out=ob, ctx=ctx)
        elif n == 512:
            call = lambda: K.fused_quantized_linear_residual(xn, *w, False, r1, r1, out=ob, ctx=ctx)
        else:
            call = lambda: K.fused_quantized_linear(xn, *w, False, out=ob, ctx=ctx)
        for _ in range(3):
            call()
        ctx.sync()
        dbg.zero_()
        torch.cuda.synchronize()
        os.environ["LELE_HIP_WHOLEK_STAMPS"] = hex(dbg.data_ptr())
        os.environ["LELE_HIP_WHOLEK_STAMPS_EM"] = str(em)
        call()
        ctx.sync()
        del os.environ["LELE_HIP_WHOLEK_STAMPS"]
        t = dbg.cpu().numpy().astype(np.float64)
        t = t[t[:, 0] > 0]
        t[t == 0] = np.nan
        d = np.diff(t, axis=1)   # per workgroup: cycles between consecutive stamps (counters are per XCD: never compare across workgroups)
        ctx.timer_start()
        for _ in range(20):
            call()
        us = ctx.timer_stop() * 1e3 / 20
        print(name, "-- whole op %.1f us --" % us, "workgroups", len(t), "-- median cycles per phase (max in brackets)")
        line = []
        for i in range(d.shape[1]):
            col = d[:, i]
            if np.isnan(col).all():
                break
            line.append("%s %.0f [%.0f]" % (names[i + 1], np.nanmedian(col), np.nanmax(col)))
        print("   " + "; ".join(line))
        print("   whole life: median %.0f cycles, max %.0f" % (np.nanmedian(np.nanmax(t, axis=1) - t[:, 0]), np.nanmax(np.nanmax(t, axis=1) - t[:, 0])))


if __name__ == "__main__":
    main()
