#!/usr/bin/env python3
"""Times every distinct convolution geometry of the Yolo-shaped network (tools/yolo_graph.py) at one batch size under the CURRENT
environment (the LELE_HIP_CONV_* switches exist in the developer's build, LELE_HIP_LAB=1, and are read once per process), HIP events around `iters` back-to-back calls.  Run it once per
variant inside ONE gpurun call and compare the files: boxes differ by 10 % from call to call, so A/B across calls says nothing.

    LELE_HIP_LAB=1 python tools/conv_ab.py --out gpurun_out/a.json;  LELE_HIP_LAB=1 LELE_HIP_CONV_TILE=rows python tools/conv_ab.py --out gpurun_out/b.json
    python tools/conv_ab.py --compare gpurun_out/a.json gpurun_out/b.json"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

# (c, oc, k, stride, out_h) of the Yolo-shaped network's group-1 convolutions (square maps; pad k // 2)
GEOMS = [
    (64, 64, 3, 1, 160), (64, 64, 3, 1, 80), (64, 32, 3, 1, 80), (32, 16, 3, 1, 80), (32, 32, 3, 1, 80), (16, 32, 3, 1, 80),
    (128, 64, 3, 1, 40), (64, 64, 3, 1, 40), (128, 32, 3, 1, 40), (64, 32, 3, 1, 40), (32, 64, 3, 1, 40), (32, 32, 3, 1, 40),
    (256, 64, 3, 1, 20), (256, 32, 3, 1, 20), (64, 64, 3, 1, 20), (128, 128, 3, 1, 20), (32, 16, 3, 1, 80), (64, 16, 3, 1, 40), (32, 8, 3, 1, 80),
    (16, 32, 3, 2, 160), (64, 64, 3, 2, 80), (128, 128, 3, 2, 40), (128, 256, 3, 2, 20), (128, 128, 3, 2, 20), (64, 64, 3, 2, 40),
    (96, 128, 1, 1, 80), (48, 64, 1, 1, 160), (64, 32, 1, 1, 160), (32, 32, 1, 1, 160), (256, 64, 1, 1, 80), (80, 80, 1, 1, 80),
    (64, 80, 1, 1, 80), (96, 64, 1, 1, 80), (64, 64, 1, 1, 80), (384, 128, 1, 1, 40), (192, 128, 1, 1, 40), (128, 80, 1, 1, 40),
    (64, 64, 1, 1, 40), (64, 32, 1, 1, 40), (512, 256, 1, 1, 20), (384, 256, 1, 1, 20), (256, 256, 1, 1, 20), (128, 256, 1, 1, 20),
    (128, 128, 1, 1, 20), (256, 80, 1, 1, 20), (128, 64, 1, 1, 20),
    # the direct kernel's layers of the reference graph (few channels)
    (3, 16, 3, 2, 320), (16, 8, 3, 1, 160), (8, 16, 3, 1, 160), (16, 16, 3, 1, 80), (16, 16, 3, 1, 20),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--compare", nargs=2, default=None)
    ap.add_argument("--act", default="silu", choices=["silu", "none", "relu"])
    ap.add_argument("--dw", action="store_true", help="the depthwise 3 x 3 layers instead (group = channels)")
    ap.add_argument("--only", default="", help="substring of the geometry label, e.g. 'k1 ' or '@160'")
    ap.add_argument("--res", action="store_true", help="conv2d_res (a residual added behind the activation) on the stride-1 geometries")
    a = ap.parse_args()
    if a.compare:
        A, B = (json.load(open(f)) for f in a.compare)
        ta = tb = 0.0
        for ra, rb in zip(A["rows"], B["rows"]):
            assert ra["geom"] == rb["geom"]
            ta, tb = ta + ra["us"], tb + rb["us"]
            print("%-28s %8.1f %8.1f us  %+6.1f %%" % (ra["geom"], ra["us"], rb["us"], 100.0 * (rb["us"] / ra["us"] - 1.0)))
        print("%-28s %8.1f %8.1f us  %+6.1f %%" % ("sum", ta, tb, 100.0 * (tb / ta - 1.0)))
        return
    import lele_amd
    from lele_amd import kernels as K
    ctx = lele_amd._lib.Ctx(0)
    rng = np.random.default_rng(7)
    rows = []
    geoms = [(80, 80, 3, 1, 80), (64, 64, 3, 1, 80), (128, 128, 3, 1, 40), (256, 256, 3, 1, 20), (32, 32, 3, 1, 160)] if a.dw else GEOMS
    for c, oc, k, s, oh in geoms:
        if a.only and a.only not in "%d->%d k%d s%d @%d" % (c, oc, k, s, oh):
            continue
        ih = oh * s
        xt = ctx.buf().upload((rng.standard_normal((a.batch, c, ih, ih))).astype(np.float32))
        from lele_amd._lib import Weight
        grp = c if a.dw else 1
        w = Weight((rng.standard_normal((oc, c // grp, k, k)) * 0.1).astype(np.float32))
        b = Weight(rng.standard_normal(oc).astype(np.float32))
        out = ctx.buf()
        conv = {"silu": K.conv2d_silu, "none": K.conv2d, "relu": lambda *p, **kw: K.conv2d_fused(*p, relu=True, **kw)}[a.act]
        fn = lambda: conv(xt, w, b, [1, 1], grp, [k // 2] * 4, [s, s], out=out, ctx=ctx)
        if a.res:
            if s != 1:
                continue
            rt = ctx.buf().upload(rng.standard_normal((a.batch, oc, oh, oh)).astype(np.float32))
            fn = lambda: K.conv2d_res(xt, w, b, rt, [1, 1], grp, [k // 2] * 4, [s, s], act={"silu": 2, "none": 0, "relu": 1}[a.act], out=out, ctx=ctx)
        for _ in range(3):
            fn()
        ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_start()
            for _ in range(a.iters):
                fn()
            best = min(best, ctx.timer_stop() / a.iters)
        flop = 2.0 * a.batch * oc * (c // grp) * k * k * oh * oh
        rows.append({"geom": "%d->%d k%d s%d @%d" % (c, oc, k, s, oh), "us": round(best * 1e3, 1), "tflops": round(flop / best / 1e9, 1)})
        print(rows[-1], flush=True)
    rec = {"batch": a.batch, "env": {k: v for k, v in os.environ.items() if k.startswith("LELE_HIP_")}, "rows": rows,
           "sum_us": round(sum(r["us"] for r in rows), 1)}
    if a.out:
        json.dump(rec, open(a.out, "w"), indent=1)
    print("sum", rec["sum_us"], "us")


if __name__ == "__main__":
    main()
