#!/usr/bin/env python3
"""conv2d_res against conv2d_silu + add, bit for bit, over the stride-1 geometries of tools/conv_ab.py (batch 8)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from conv_ab import GEOMS
import lele_amd
from lele_amd import kernels as K
from lele_amd._lib import Weight
ctx = lele_amd._lib.Ctx(0)
rng = np.random.default_rng(3)
bad = 0
for n in (8, 64):
    for c, oc, k, s, oh in GEOMS:
        if s != 1 or (n == 64 and oh > 80):
            continue
        x = ctx.buf().upload(rng.standard_normal((n, c, oh, oh)).astype(np.float32))
        r = ctx.buf().upload(rng.standard_normal((n, oc, oh, oh)).astype(np.float32))
        w = Weight((rng.standard_normal((oc, c, k, k)) * 0.1).astype(np.float32))
        b = Weight(rng.standard_normal(oc).astype(np.float32))
        y = K.conv2d_silu(x, w, b, [1, 1], 1, [k // 2] * 4, [1, 1], out=ctx.buf(), ctx=ctx)
        want = K.add(y, r, out=ctx.buf(), ctx=ctx).numpy()
        got = K.conv2d_res(x, w, b, r, [1, 1], 1, [k // 2] * 4, [1, 1], act=2, out=ctx.buf(), ctx=ctx).numpy()
        same = np.array_equal(want, got)
        if not same:
            bad += 1
            d = np.argwhere(want != got)
            print("MISMATCH n=%d %d->%d k%d @%d: %d elements, first %s, max |d| %.3g" % (n, c, oc, k, oh, len(d), d[0].tolist(), np.abs(want - got).max()), flush=True)
print("bad", bad)
