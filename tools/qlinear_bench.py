#!/usr/bin/env python3
"""Per-call time of fused_quantized_linear at the SenseVoice shapes, free of host overhead: 20 calls are recorded into one
hipGraph and the replay is timed with HIP events on the ctx stream.  Variants: the default routing (register-stationary kernels
where the sizes allow) against the tiled chain (LELE_HIP_IGEMM_RS=0); with the lab library (LELE_HIP_LAB=1) also the tiled
kernel's tile shapes (LELE_HIP_IGEMM_TILE, a lab-only switch).

    gpurun -- 'python tools/qlinear_bench.py --out gpurun_out/qlinear.json'
"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--calls", type=int, default=20)
    ap.add_argument("--only", default="")
    ap.add_argument("--compute-bound", action="store_true",
                    help="VERDICT r5 item 6: one slice of M rows, M in {8192, 32768}, K in {512, 2048, 4096}, N in {2048, 4096}: what "
                         "the i8 cores reach when K is not 512 (fused op with its dynamic quantisation, and mat_mul_integer alone)")
    ap.add_argument("--shapes", default="", help="with --compute-bound: only these MxKxN (comma-separated), e.g. 8192x4096x4096")
    args = ap.parse_args()
    if args.compute_bound:
        return compute_bound(args)
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    ctx = lele_amd._lib.Ctx(0)
    rng = np.random.default_rng(0)
    shapes = [("c4 qkv", 32, 171, 512, 1536, False), ("c4 out", 32, 171, 512, 512, False), ("c4 ffn1", 32, 171, 512, 2048, True),
              ("c4 ffn2", 32, 171, 2048, 512, False), ("c3 qkv", 1, 504, 512, 1536, False), ("c3 out", 1, 504, 512, 512, False),
              ("c3 ffn1", 1, 504, 512, 2048, True), ("c3 ffn2", 1, 504, 2048, 512, False)]
    variants = [("default", {}), ("tiled chain", {"LELE_HIP_IGEMM_RS": "0"})]
    if os.environ.get("LELE_HIP_LAB") == "1":
        variants += [("tiled chain tile=%d" % t, {"LELE_HIP_IGEMM_RS": "0", "LELE_HIP_IGEMM_TILE": str(t)}) for t in range(1, 15)]
    res = []
    for name, b, m, k, n, relu in shapes:
        if args.only and args.only not in name:
            continue
        x = ctx.buf().upload(rng.standard_normal((b, m, k)).astype(np.float32))
        g, be = Weight(np.ones(k, np.float32)), Weight(np.zeros(k, np.float32))
        xn = K.layer_norm(x, g, be, -1, 1e-5, out=ctx.buf(), ctx=ctx)     # leaves row statistics, as in the model
        w = (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)),
             Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
             Weight((rng.standard_normal(n) * 0.02).astype(np.float32)))
        ob = ctx.buf()
        row = {"shape": name, "rows": b * m, "k": k, "n": n}
        ref = None
        for vname, env in variants:
            old = {kk: os.environ.get(kk) for kk in env}
            os.environ.update(env)
            try:
                out = K.fused_quantized_linear(xn, *w, relu, out=ob, ctx=ctx)
                got = out.numpy().copy()
                if ref is None:
                    ref = got
                same = bool(np.array_equal(got, ref))
                ctx.sync()
                ctx.graph_begin()
                for _ in range(args.calls):
                    K.fused_quantized_linear(xn, *w, relu, out=ob, ctx=ctx)
                gr = ctx.graph_end()
                gr.launch()
                ctx.sync()
                ctx.timer_start()
                for _ in range(10):
                    gr.launch()
                us = ctx.timer_stop() * 1e3 / (10 * args.calls)
                gr.close()
            finally:
                for kk, v in old.items():
                    if v is None:
                        os.environ.pop(kk, None)
                    else:
                        os.environ[kk] = v
            byts = 4 * b * m * k + k * n + 8 * n + 4 * b * m * n
            row[vname] = {"us": round(us, 2), "same_bits": same, "hbm_gbs": round(byts / us / 1e3, 1),
                          "tops": round(2 * b * m * k * n / us / 1e6, 1)}
        print(json.dumps(row), flush=True)
        res.append(row)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(res, open(args.out, "w"), indent=1)


def _timed(ctx, fn, calls, graph=True):
    """us per call: `calls` calls recorded into one hipGraph and replayed (no host overhead); eager trains of calls when the op's scratch
    does not fit the context's staging arena (an op that allocates cannot be captured) -- those calls take >= 100 us each"""
    fn()
    ctx.sync()
    if not graph:
        ctx.timer_start()
        for _ in range(5 * calls):
            fn()
        return ctx.timer_stop() * 1e3 / (5 * calls)
    ctx.graph_begin()
    for _ in range(calls):
        fn()
    gr = ctx.graph_end()
    gr.launch()
    ctx.sync()
    ctx.timer_start()
    for _ in range(5):
        gr.launch()
    us = ctx.timer_stop() * 1e3 / (5 * calls)
    gr.close()
    return us


def compute_bound(args):
    """TOP/s against the 3944 TOP/s dense i8 peak.  `fused` = lele's fused_quantized_linear (range + quantisation + GEMM + f32
    epilogue: the op a model pays for), default routing and the tiled chain; `mmi` = mat_mul_integer on u8 codes that are already
    quantised (the GEMM kernel + its packing of the activation, no range pass): the nearest thing to the core alone."""
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    ctx = lele_amd._lib.Ctx(0)
    rng = np.random.default_rng(0)
    PEAK = 3944.0
    res = []
    grid = [(m, k, n) for m in (8192, 32768) for k in (512, 2048, 4096) for n in (2048, 4096)]
    if args.shapes:   # any MxKxN, not only the grid's
        grid = [tuple(int(v) for v in s_.split("x")) for s_ in args.shapes.split(",")]
    for m, k, n in grid:
        for _once in (0,):
            for _once2 in (0,):
                x = ctx.buf().upload(rng.standard_normal((1, m, k)).astype(np.float32))
                wq = np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)
                w = (Weight(wq), Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)),
                     Weight(np.array([128.0], np.float32)), Weight((rng.standard_normal(n) * 0.02).astype(np.float32)))
                ob = ctx.buf()
                graph = m * k <= 32 << 20     # the quantised activation has to fit the 64 MiB staging arena for the op to be captured
                row = {"rows": m, "k": k, "n": n, "GOP": round(2 * m * k * n / 1e9, 1), "timed": "graph replay" if graph else "eager train"}
                ref = None
                for vname, env in (("fused default", {}), ("fused tiled chain", {"LELE_HIP_IGEMM_RS": "0"})):
                    old = {kk: os.environ.get(kk) for kk in env}
                    os.environ.update(env)
                    try:
                        got = K.fused_quantized_linear(x, *w, False, out=ob, ctx=ctx).numpy().copy()
                        if ref is None:
                            ref = got
                        us = _timed(ctx, lambda: K.fused_quantized_linear(x, *w, False, out=ob, ctx=ctx), args.calls, graph)
                    finally:
                        for kk, v in old.items():
                            os.environ.pop(kk, None) if v is None else os.environ.__setitem__(kk, v)
                    tops = 2 * m * k * n / us / 1e6
                    row[vname] = {"us": round(us, 2), "same_bits": bool(np.array_equal(got, ref)), "tops": round(tops, 1),
                                  "frac_of_i8_peak": round(tops / PEAK, 4)}
                a = ctx.buf().upload(rng.integers(0, 256, (1, m, k)).astype(np.float32))
                azp, bzp = Weight(np.array([128.0], np.float32)), Weight(np.array([127.0], np.float32))
                wb = Weight(wq)
                us = _timed(ctx, lambda: K.mat_mul_integer(a, wb, azp, bzp, out=ob, ctx=ctx), args.calls, graph)
                tops = 2 * m * k * n / us / 1e6
                row["mmi"] = {"us": round(us, 2), "tops": round(tops, 1), "frac_of_i8_peak": round(tops / PEAK, 4)}
                print(json.dumps(row), flush=True)
                res.append(row)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump({"peak_i8_tops": PEAK, "rows": res}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
