#!/usr/bin/env python3
"""Round 6: the fused half-layer forms against the call sequences they replace, at the SenseVoice shapes -- same bits required, time
per call from hipGraph replays of 20 calls (HIP events on the ctx stream).

    out+ln      fused_quantized_linear_residual_ln            vs  fused_quantized_linear_residual -> layer_norm
    ffn+ln      fused_ffn_quantized_ln                        vs  fused_ffn_quantized -> layer_norm
    sanm+ln     sanm_out_block                                vs  depthwise_conv1d_tlc -> fused_quantized_linear_residual -> layer_norm

    gpurun -- 'python tools/block_bench.py --out gpurun_out/block_bench.json'
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.rs_bench import env, timed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--only", default="")
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

    def ln():
        return Weight((1 + 0.1 * rng.standard_normal(512)).astype(np.float32)), Weight((0.1 * rng.standard_normal(512)).astype(np.float32))
    res = []
    for name, b, m in (("c4", 32, 171), ("c3", 1, 504), ("64x171", 64, 171)):
        x = ctx.buf().upload((rng.standard_normal((b, m, 512)) * rng.uniform(0.5, 2.0, (b, 1, 1))).astype(np.float32))
        g0, b0 = ln()
        xn = K.layer_norm(x, g0, b0, -1, 1e-5, out=ctx.buf(), ctx=ctx)   # leaves row statistics, as in the model
        r1 = ctx.buf().upload(rng.standard_normal((b, m, 512)).astype(np.float32))
        r2 = ctx.buf().upload(rng.standard_normal((b, m, 512)).astype(np.float32))
        g1, b1 = ln()
        wo, w1, w2 = lin(512, 512), lin(512, 2048), lin(2048, 512)
        o = [ctx.buf() for _ in range(4)]
        cases = {}
        if hasattr(K, "fused_quantized_linear_residual_ln"):
            def seq_out():
                x1 = K.fused_quantized_linear_residual(xn, *wo, False, r1, r2, out=o[0], ctx=ctx)
                return x1, K.layer_norm(x1, g1, b1, -1, 1e-5, out=o[1], ctx=ctx)

            def fus_out():
                return K.fused_quantized_linear_residual_ln(xn, *wo, False, r1, r2, g1, b1, 1e-5, outs=[o[2], o[3]], ctx=ctx)
            cases["out+ln"] = (seq_out, fus_out)
        if hasattr(K, "fused_ffn_quantized_ln"):
            def seq_ffn():
                y = K.fused_ffn_quantized(xn, *w1, *w2, False, r1, out=o[0], ctx=ctx)
                return y, K.layer_norm(y, g1, b1, -1, 1e-5, out=o[1], ctx=ctx)

            def fus_ffn():
                return K.fused_ffn_quantized_ln(xn, *w1, *w2, False, r1, None, g1, b1, 1e-5, outs=[o[2], o[3]], ctx=ctx)
            cases["ffn+ln"] = (seq_ffn, fus_ffn)
        if hasattr(K, "sanm_out_block"):
            qkv = ctx.buf().upload(rng.standard_normal((b, m, 1536)).astype(np.float32))
            fw = Weight((rng.standard_normal((512, 1, 11)) / np.sqrt(11)).astype(np.float32))
            mb = ctx.buf()

            def seq_sanm():
                mem = K.depthwise_conv1d_tlc(qkv, fw, None, 5, 5, False, 1024, True, out=mb, ctx=ctx)
                x1 = K.fused_quantized_linear_residual(xn, *wo, False, mem, r2, out=o[0], ctx=ctx)
                return x1, K.layer_norm(x1, g1, b1, -1, 1e-5, out=o[1], ctx=ctx)

            def fus_sanm():
                return K.sanm_out_block(xn, *wo, False, qkv, fw, None, 1024, 5, 5, r2, g1, b1, 1e-5, outs=[o[2], o[3]], ctx=ctx)
            cases["sanm+ln"] = (seq_sanm, fus_sanm)
        for cname, (seq, fus) in cases.items():
            if args.only and args.only not in cname:
                continue
            a = [t.numpy().copy() for t in seq()]
            c = [t.numpy().copy() for t in fus()]
            same = all(np.array_equal(p, q) for p, q in zip(a, c))
            # what the NEXT quantised linear makes of the normalised result (its range comes from the statistics left beside it)
            nxt = [K.fused_quantized_linear(t[1], *w1, True, ctx=ctx).numpy().copy() for t in (seq(), fus())]
            row = {"case": cname, "shape": name, "rows": b * m, "same_bits": bool(same), "next_linear_same_bits": bool(np.array_equal(*nxt)),
                   "sequence_us": round(timed(ctx, seq), 2), "fused_us": round(timed(ctx, fus), 2)}
            print(json.dumps(row), flush=True)
            res.append(row)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
