#!/usr/bin/env python3
"""A/B of the register-stationary i8 GEMM route (igemm_rs.h, LELE_HIP_IGEMM_RS=1, the default) against the tiled kernels
(LELE_HIP_IGEMM_RS=0) at the SenseVoice shapes: same bits required, per-call time from hipGraph replays of 20 calls.

    gpurun -- 'python tools/rs_bench.py --out gpurun_out/rs_bench.json'
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def timed(ctx, fn, calls=20, reps=10):
    fn()
    ctx.sync()
    ctx.graph_begin()
    for _ in range(calls):
        fn()
    gr = ctx.graph_end()
    gr.launch()
    ctx.sync()
    ctx.timer_start()
    for _ in range(reps):
        gr.launch()
    us = ctx.timer_stop() * 1e3 / (reps * calls)
    gr.close()
    return us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default="")
    ap.add_argument("--rs-min", default="")
    args = ap.parse_args()
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    ctx = lele_amd._lib.Ctx(0)
    rng = np.random.default_rng(0)

    def lin(k, n):
        return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)),
                Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
                Weight((rng.standard_normal(n) * 0.02).astype(np.float32)))
    shapes = [("c4 qkv", 32, 171, 512, 1536, 0), ("c4 out+res2", 32, 171, 512, 512, 2), ("c4 ffn1 relu", 32, 171, 512, 2048, 0),
              ("c4 ffn2+res1", 32, 171, 2048, 512, 1), ("c3 qkv", 1, 504, 512, 1536, 0), ("c3 out+res2", 1, 504, 512, 512, 2),
              ("c3 ffn1 relu", 1, 504, 512, 2048, 0), ("c3 ffn2+res1", 1, 504, 2048, 512, 1)]
    variants = [("rs", {"LELE_HIP_IGEMM_RS": 1}), ("tiled", {"LELE_HIP_IGEMM_RS": 0})]
    if args.rs_min:
        variants.insert(1, ("rs min=%s" % args.rs_min, {"LELE_HIP_IGEMM_RS": 1, "LELE_HIP_IGEMM_RS_MIN": args.rs_min}))
    res = []
    for name, b, m, k, n, nres in shapes:
        if args.only and args.only not in name:
            continue
        x = ctx.buf().upload((rng.standard_normal((b, m, k)) * rng.uniform(0.5, 2.0, (b, 1, 1))).astype(np.float32))
        g, be = Weight(np.ones(k, np.float32)), Weight(np.zeros(k, np.float32))
        xn = K.layer_norm(x, g, be, -1, 1e-5, out=ctx.buf(), ctx=ctx)  # leaves row statistics, as in the model
        w = lin(k, n)
        relu = "relu" in name
        r1 = ctx.buf().upload(rng.standard_normal((b, m, n)).astype(np.float32))
        r2 = ctx.buf().upload(rng.standard_normal((b, m, n)).astype(np.float32))
        ob = ctx.buf()

        def call():
            if nres == 0:
                return K.fused_quantized_linear(xn, *w, relu, out=ob, ctx=ctx)
            if nres == 1:
                return K.fused_quantized_linear_residual(xn, *w, relu, r1, out=ob, ctx=ctx)
            return K.fused_quantized_linear_residual(xn, *w, relu, r1, r2, out=ob, ctx=ctx)
        row = {"shape": name, "rows": b * m, "k": k, "n": n}
        ref = None
        byts = 4 * b * m * k + k * n + 8 * n + 4 * b * m * n * (1 + nres)
        for vname, e in variants:
            with env(**e):
                got = call().numpy().copy()
                ref = got if ref is None else ref
                us = timed(ctx, call)
            row[vname] = {"us": round(us, 2), "same_bits": bool(np.array_equal(got, ref)), "hbm_gbs": round(byts / us / 1e3, 1),
                          "hbm_frac": round(byts / us / 1e3 / 8000, 3), "tops": round(2 * b * m * k * n / us / 1e6, 1)}
        print(json.dumps(row), flush=True)
        res.append(row)
    # the feed-forward block as one call (hidden layer never stored)
    for name, b, m in (("c4 ffn block", 32, 171), ("c3 ffn block", 1, 504)):
        if args.only and args.only not in name:
            continue
        x = ctx.buf().upload((rng.standard_normal((b, m, 512)) * rng.uniform(0.5, 2.0, (b, 1, 1))).astype(np.float32))
        g, be = Weight(np.ones(512, np.float32)), Weight(np.zeros(512, np.float32))
        xn = K.layer_norm(x, g, be, -1, 1e-5, out=ctx.buf(), ctx=ctx)
        w1, w2 = lin(512, 2048), lin(2048, 512)
        r1 = ctx.buf().upload(rng.standard_normal((b, m, 512)).astype(np.float32))
        ob = ctx.buf()

        def call():
            return K.fused_ffn_quantized(xn, *w1, *w2, False, r1, out=ob, ctx=ctx)
        row = {"shape": name, "rows": b * m}
        ref = None
        for vname, e in variants:
            with env(**e):
                got = call().numpy().copy()
                ref = got if ref is None else ref
                us = timed(ctx, call)
            row[vname] = {"us": round(us, 2), "same_bits": bool(np.array_equal(got, ref))}
        print(json.dumps(row), flush=True)
        res.append(row)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
