#!/usr/bin/env python3
"""Determinism stress of lele_hip_fused_ffn_quantized at the configs[3] shard shape: the same call repeated, every result compared
with the first (and with the tiled-kernel route)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from lele_amd import kernels as K
    from lele_amd._lib import Ctx, Weight
    ctx = Ctx()
    rng = np.random.default_rng(0)
    b, m, k, n = 32, 171, 512, 2048

    def lin(kk, nn):
        return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((kk, nn))), 0, 255).astype(np.float32)),
                Weight((np.abs(rng.standard_normal(nn)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
                Weight((rng.standard_normal(nn) * 0.02).astype(np.float32)))
    w, w2 = lin(k, n), lin(n, 512)
    x = ctx.buf().upload((rng.standard_normal((b, m, k)) * rng.uniform(0.3, 3, (b, 1, 1))).astype(np.float32))
    os.environ["LELE_HIP_IGEMM_RS"] = "0"
    ref = K.fused_ffn_quantized(x, *w, *w2, False, ctx=ctx).numpy().copy()
    del os.environ["LELE_HIP_IGEMM_RS"]
    bad = 0
    ob = ctx.buf()
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
        got = K.fused_ffn_quantized(x, *w, *w2, False, out=ob, ctx=ctx).numpy()
        if not np.array_equal(got, ref):
            d = (got != ref).reshape(b, -1).any(axis=1)
            bad += 1
            if bad <= 5:
                print("iteration", it, "utterances that differ:", np.nonzero(d)[0].tolist())
    print("mismatching iterations:", bad)


if __name__ == "__main__":
    main()
