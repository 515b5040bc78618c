#!/usr/bin/env python3
"""conv_integer_from_f32 at Yolo sizes: the i8 matrix-core route against lele's own formulation (centre in f32, f32 convolution), which the
developer's build still takes under LELE_HIP_CONV_INTEGER_F32=1.  One JSON object; HIP events around `iters` calls.

    LELE_HIP_LAB=1 python tools/conv_integer_bench.py > a.json;  LELE_HIP_LAB=1 LELE_HIP_CONV_INTEGER_F32=1 python tools/conv_integer_bench.py > b.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    ctx = lele_amd._lib.Ctx(0)
    rng = np.random.default_rng(3)
    rows = []
    for n, c, h, oc, k, s in ((1, 64, 80, 64, 3, 1), (1, 128, 40, 128, 3, 1), (1, 256, 20, 256, 3, 1), (1, 128, 80, 128, 1, 1), (1, 64, 160, 64, 3, 2),
                              (16, 64, 80, 64, 3, 1), (16, 128, 40, 256, 1, 1), (64, 64, 80, 64, 3, 1), (64, 128, 40, 128, 3, 1), (64, 96, 80, 128, 1, 1)):
        x = ctx.buf().upload((rng.standard_normal((n, c, h, h)) * 2).astype(np.float32))
        w = Weight(rng.integers(0, 256, (oc, c, k, k)).astype(np.float32))
        zw = np.array([128.0], np.float32)
        fn = lambda: K.conv_integer_from_f32(x, w, zw, [1, 1], 1, [k // 2] * 4, [s, s], ctx=ctx)
        for _ in range(3):
            fn()
        ctx.sync()
        best = 1e9
        for _ in range(3):
            ctx.timer_start()
            for _ in range(10):
                fn()
            best = min(best, ctx.timer_stop() / 10)
        oh = h // s
        gop = 2.0 * n * oc * c * k * k * oh * oh / 1e9
        rows.append({"shape": "%dx %d->%d k%d s%d @%d" % (n, c, oc, k, s, oh), "us": round(best * 1e3, 1), "TOP/s": round(gop / best, 2)})
    print(json.dumps({"route": "f32" if os.environ.get("LELE_HIP_CONV_INTEGER_F32") else "i8", "rows": rows}))


if __name__ == "__main__":
    main()
