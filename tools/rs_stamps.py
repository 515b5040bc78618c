#!/usr/bin/env python3
"""Phase timing of igemm_rs_kernel from the inside (lab build only: LELE_HIP_LAB=1 python -m lele_amd.build, then
LELE_HIP_LAB=1 python tools/rs_stamps.py).  Consumer wave 0 and the loader of every workgroup stamp the 100 MHz wall clock at
their phase boundaries (LELE_HIP_RS_STAMPS = device address of a [grid][9][32] i64 buffer: every wave, wave 8 = the loader); printed: median / max over
workgroups of every stamp relative to the EARLIEST entry stamp of the launch, in microseconds."""
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
    dbg = torch.zeros((256, 9, 32), dtype=torch.int64, device="cuda")

    def lin(kk, nn):
        return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((kk, nn))), 0, 255).astype(np.float32)),
                Weight((np.abs(rng.standard_normal(nn)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
                Weight((rng.standard_normal(nn) * 0.02).astype(np.float32)))
    for name, n, em in (("qkv (EM 0)", 1536, 0), ("out + 2 residuals (EM 0)", 512, 0), ("ffn hidden, range pass (EM 1)", 2048, 1),
                        ("ffn hidden, quantise pass (EM 2)", 2048, 2), ("ffn hidden, both passes in one launch (EM 3)", 2048, 3)):
        b, m, k = 32, 171, 512
        x = ctx.buf().upload(rng.standard_normal((b, m, k)).astype(np.float32))
        g, be = Weight(np.ones(k, np.float32)), Weight(np.zeros(k, np.float32))
        xn = K.layer_norm(x, g, be, -1, 1e-5, out=ctx.buf(), ctx=ctx)
        w, w2 = lin(k, n), lin(n, 512)
        r1 = ctx.buf().upload(rng.standard_normal((b, m, n)).astype(np.float32))
        ob = ctx.buf()
        if em:
            call = lambda: K.fused_ffn_quantized(xn, *w, *w2, False, out=ob, ctx=ctx)
        elif n == 512:
            call = lambda: K.fused_quantized_linear_residual(xn, *w, False, r1, r1, out=ob, ctx=ctx)
        else:
            call = lambda: K.fused_quantized_linear(xn, *w, False, out=ob, ctx=ctx)
        os.environ["LELE_HIP_FFN_ONE_LAUNCH"] = "2" if em == 3 else "0"
        for _ in range(3):
            call()
        ctx.sync()
        dbg.zero_()
        torch.cuda.synchronize()
        os.environ["LELE_HIP_RS_STAMPS"] = hex(dbg.data_ptr())
        os.environ["LELE_HIP_RS_STAMPS_EM"] = str(em)
        call()
        ctx.sync()
        del os.environ["LELE_HIP_RS_STAMPS"]
        t = dbg.cpu().numpy().astype(np.float64)
        cyc = t[:, 0, 31].copy()
        t[:, :, 31] = 0
        t[t == 0] = np.nan
        span = np.nanmax(t[:, 0, :], axis=1) - t[:, 0, 0]   # wave 0: entry -> last stamp, in 10 ns ticks
        print("   shader clock over the consumer's life: median %.0f MHz" % np.nanmedian(cyc / (span / 100.0)))
        t0 = np.nanmin(t[:, :, 0])
        t = (t - t0) / 100.0   # us since the first workgroup started
        print("==", name, "-- workgroups with stamps:", int(np.isfinite(t[:, 0, 0]).sum()))
        for who, label in [(w, "consumer wave %d: entry, then per tile {past the barrier, products done}, end" % w) for w in range(8)] + [
                           (8, "loader: entry, ring primed, then per tile {landed, consumers arrived}")]:
            med, mx = np.nanmedian(t[:, who, :], axis=0), np.nanmax(t[:, who, :], axis=0)
            keep = np.isfinite(med)
            print("  " + label)
            print("    median us:", " ".join("%.2f" % v for v in med[keep]))
            print("    max    us:", " ".join("%.2f" % v for v in mx[keep]))


if __name__ == "__main__":
    main()
