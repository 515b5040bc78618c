#!/usr/bin/env python3
"""Phase timing of the stride-1 window convolution from inside (lab build: LELE_HIP_LAB=1; lab switch LELE_HIP_CONV_STAMPS = device
address of an i64 buffer [workgroups][8 waves][64]): the shader clock at the phase boundaries of the first items of every workgroup.
Prints median cycles (differences inside a wave only: the counters are per XCD).

    LELE_HIP_LAB=1 python tools/conv_stamps.py --geom 64,64,3,160 [--geom 48,64,1,160]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--geom", action="append", default=[], help="c,oc,k,out_h")
    ap.add_argument("--batch", type=int, default=64)
    a = ap.parse_args()
    import torch
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    ctx = lele_amd._lib.Ctx(0)
    rng = np.random.default_rng(7)
    for gs in a.geom or ["64,64,3,160"]:
        c, oc, k, oh = (int(v) for v in gs.split(","))
        xt = ctx.buf().upload(rng.standard_normal((a.batch, c, oh, oh)).astype(np.float32))
        w = Weight((rng.standard_normal((oc, c, k, k)) * 0.1).astype(np.float32))
        b = Weight(rng.standard_normal(oc).astype(np.float32))
        out = ctx.buf()
        fn = lambda: K.conv2d_silu(xt, w, b, [1, 1], 1, [k // 2] * 4, [1, 1], out=out, ctx=ctx)
        for _ in range(3):
            fn()
        ctx.sync()
        nwg = 512
        dbg = torch.zeros((nwg, 8, 64), dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        os.environ["LELE_HIP_CONV_STAMPS"] = hex(dbg.data_ptr())
        fn()
        ctx.sync()
        del os.environ["LELE_HIP_CONV_STAMPS"]
        ctx.timer_start()
        for _ in range(10):
            fn()
        us = ctx.timer_stop() * 1e2
        t = dbg.cpu().numpy().astype(np.float64)
        nchunk = c // 16
        per_item = 2 * nchunk + 2 + 5   # + e0 .. e4 inside the epilogue
        cons = t[:, 0, :]            # consumer wave 0
        prod = t[:, 4, :]            # loader wave 0
        live = cons[:, 1 + 2 * per_item] > 0
        cons, prod = cons[live], prod[live]
        d = lambda x: np.median(x)
        print("%d -> %d k%d @%d x %d images: %.1f us a call (stamped build), %d workgroups with >= 2 items; median cycles" % (c, oc, k, oh, a.batch, us, int(live.sum())))
        print("  consumer wave 0: entry -> first chunk parked (B_0) %d" % d(cons[:, 1] - cons[:, 0]))
        for it in range(2):
            base = 2 + it * per_item
            mult = [cons[:, base + 2 * q] - cons[:, base + 2 * q - 1] for q in range(nchunk)]
            wait = [cons[:, base + 2 * q + 1] - cons[:, base + 2 * q] for q in range(nchunk)]
            e = [cons[:, base + 2 * nchunk + k] - cons[:, base + 2 * nchunk + k - 1] for k in range(6)]   # -> e0, e1, e2, e3, e4, out
            epi = cons[:, base + 2 * nchunk + 5] - cons[:, base + 2 * nchunk - 1]
            ebar = cons[:, base + 2 * nchunk + 6] - cons[:, base + 2 * nchunk + 5]
            tot = cons[:, base + 2 * nchunk + 6] - cons[:, base - 1]
            print("      epilogue: entry %d | bias requested + coordinates %d | LDS turn issued %d | strip 0 out %d | strips 1.. out %d | return %d" % tuple(int(d(x)) for x in e))
            print("    item %d: products issued per chunk %s | barrier waits %s | strips (LDS turn, bias, activation, stores issued) %d | E barrier %d | item total %d"
                  % (it, [int(d(m)) for m in mult], [int(d(w_)) for w_ in wait], d(epi), d(ebar), d(tot)))
        print("  loader wave 0: entry -> D chunks requested %d -> chunk 0 parked %d -> B_0 %d" % (d(prod[:, 1] - prod[:, 0]), d(prod[:, 2] - prod[:, 1]), d(prod[:, 3] - prod[:, 2])))
        rows = []
        for q in range(1, min(2 * nchunk + 1, 19)):
            o = 4 + 3 * (q - 1)
            rows.append((q, int(d(prod[:, o] - prod[:, o - 1])), int(d(prod[:, o + 1] - prod[:, o])), int(d(prod[:, o + 2] - prod[:, o + 1]))))
        print("    chunk q: (request issued, E wait + park, B_q wait) " + " ".join("%d:(%d, %d, %d)" % r for r in rows))


if __name__ == "__main__":
    main()
