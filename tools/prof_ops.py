#!/usr/bin/env python3
"""Run a few operators in a loop so that `rocprofv3 --kernel-trace --stats` shows their per-kernel breakdown.
usage: rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o ops -- python tools/prof_ops.py [quant|matmul|conv|all]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lele_amd  # noqa: E402
from lele_amd import kernels as K  # noqa: E402
from lele_amd._lib import Weight  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "all"
ctx = lele_amd._lib.Ctx(0)
rng = np.random.default_rng(0)


def dev(a):
    return ctx.buf().upload(np.ascontiguousarray(a.astype(np.float32)))


out = ctx.buf()
if what in ("quant", "all"):
    for m, k, n in [(8064, 512, 2048), (504, 512, 2048)]:
        x = dev(rng.standard_normal((1, m, k)))
        w = Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
        ws, wz, bs = Weight(np.full(n, 0.01, np.float32)), Weight(np.array([128.0], np.float32)), Weight(np.zeros(n, np.float32))
        for _ in range(20):
            K.fused_quantized_linear(x, w, ws, wz, bs, False, out=out, ctx=ctx)
if what in ("matmul", "all"):
    a, b = dev(rng.standard_normal((4096, 4096))), dev(rng.standard_normal((4096, 4096)))
    for _ in range(10):
        K.matmul(a, b, out=out, ctx=ctx)
    a, b = dev(rng.standard_normal((4, 504, 128))), dev(rng.standard_normal((4, 128, 504)))
    for _ in range(20):
        K.matmul(a, b, out=out, ctx=ctx)
if what == "attn":  # batched short-K products of the C4 attention (64x64 tiles)
    a, b = dev(rng.standard_normal((128, 171, 128))), dev(rng.standard_normal((128, 128, 171)))
    for _ in range(10):
        K.matmul(a, b, out=out, ctx=ctx)
if what in ("conv", "all"):
    x = dev(rng.standard_normal((16, 64, 160, 160)))
    w, bs = Weight(rng.standard_normal((64, 64, 3, 3)).astype(np.float32) * 0.1), Weight(np.zeros(64, np.float32))
    for _ in range(5):
        K.conv2d_silu(x, w, bs, [1, 1], 1, [1, 1, 1, 1], [1, 1], out=out, ctx=ctx)
ctx.sync()
