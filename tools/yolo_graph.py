#!/usr/bin/env python3
"""BASELINE configs[4]: a Yolo26n-seg-SHAPED network at batch N as ONE graph.

lele's generated Yolo26n-seg file bakes N = 1 into its reshapes (examples/yolo26n-seg/src/yolo26seg.rs; SURVEY.md 8(d): 118
convolutions, 9.13 GFLOP an image, outputs [1, 300, 38] and [1, 32, 160, 160]); the ONNX it was compiled from is not in the
tree.  This builds a network of the same family from its public description -- n-scale widths (16 .. 256), stride-2 stem,
C3k2 stages, SPPF, a two-head 20 x 20 position-sensitive attention block, an FPN / PAN neck, per-level box / class / mask
branches with depth-wise 3 x 3 convolutions, a transposed-convolution prototype branch, and the NMS-free top-300 tail
(2 TopK, 3 GatherElements) -- as ONNX bytes with the batch size in the graph, pushes it through lele_amd.compiler, and runs the
compiled plan:

    python tools/yolo_graph.py --batch 64 --out gpurun_out/yolo_n64.json      # on the GPU box

Checks: every image of the batch-N forward equals the batch-1 plan's forward of that image (same weights) within 1e-4 (and
reports whether bit for bit); times the batch-N graph replay; counts the multiply-adds the convolutions perform.
Weights are synthetic (seeded), scaled so that activations stay O(1) through the depth."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class T:
    """a tensor of the graph under construction: name, channels, spatial size (square)"""

    def __init__(self, name, c, hw):
        self.name, self.c, self.hw = name, c, hw


class Builder:
    def __init__(self, batch, size=640, seed=26, classes=80, masks=32):
        from lele_amd.compiler import onnx_pb as pb
        self.pb, self.batch, self.size, self.nc, self.nm = pb, batch, size, classes, masks
        self.g = pb.Graph([], [pb.ValueInfo("images", pb.FLOAT, [batch, 3, size, size])], [])
        self.rng = np.random.default_rng(seed)
        self.count = 0
        self.macs = 0          # multiply-adds of the convolutions and matrix products, one image
        self.convs = 0

    def fresh(self, tag):
        self.count += 1
        return "%s_%d" % (tag, self.count)

    def const(self, arr, tag="c"):
        nm = self.fresh(tag)
        self.g.initializer.append(self.pb.Tensor(nm, arr))
        return nm

    def node(self, op, ins, tag, nout=1, **attrs):
        outs = [self.fresh(tag) for _ in range(nout)]
        self.g.node.append(self.pb.Node(op, list(ins), outs, **attrs))
        return outs[0] if nout == 1 else outs

    # ---- layers
    def conv(self, x, cout, k=1, s=1, act=True, groups=1):
        cin_g = x.c // groups
        fan = cin_g * k * k
        w = (self.rng.standard_normal((cout, cin_g, k, k)) * np.sqrt((2.0 if act else 1.0) / fan)).astype(np.float32)
        b = (self.rng.standard_normal(cout) * 0.05).astype(np.float32)
        y = self.node("Conv", [x.name, self.const(w, "w"), self.const(b, "b")], "conv", kernel_shape=[k, k], strides=[s, s],
                      pads=[k // 2] * 4, group=groups, dilations=[1, 1])
        hw = (x.hw + 2 * (k // 2) - k) // s + 1
        self.macs += cout * cin_g * k * k * hw * hw
        self.convs += 1
        if act:  # SiLU as the exporter writes it: Sigmoid + Mul (the compiler folds both into conv2d_silu)
            sg = self.node("Sigmoid", [y], "sig")
            y = self.node("Mul", [y, sg], "silu")
        return T(y, cout, hw)

    def add(self, a, b):
        return T(self.node("Add", [a.name, b.name], "add"), a.c, a.hw)

    def cat(self, xs):
        return T(self.node("Concat", [x.name for x in xs], "cat", axis=1), sum(x.c for x in xs), xs[0].hw)

    def split2(self, x):
        half = x.c // 2
        a, b = self.node("Split", [x.name, self.const(np.array([half, half], np.int64), "sp")], "split", nout=2, axis=1)
        return T(a, half, x.hw), T(b, half, x.hw)

    def bottleneck(self, x, shortcut=True, e=0.5):
        y = self.conv(self.conv(x, int(x.c * e), 3), x.c, 3)
        return self.add(x, y) if shortcut else y

    def c3k(self, x, cout, n=2):
        c_ = cout // 2
        a, b = self.conv(x, c_, 1), self.conv(x, c_, 1)
        for _ in range(n):
            a = self.bottleneck(a, True, 1.0)
        return self.conv(self.cat([a, b]), cout, 1)

    def c3k2(self, x, cout, c3k=False, e=0.5, shortcut=True):
        c = int(cout * e)
        y0, y1 = self.split2(self.conv(x, 2 * c, 1))
        m = self.c3k(y1, c, 2) if c3k else self.bottleneck(y1, shortcut, 0.5)
        return self.conv(self.cat([y0, y1, m]), cout, 1)

    def sppf(self, x, cout):
        y = self.conv(x, x.c // 2, 1)
        ps = [y]
        for _ in range(3):
            ps.append(T(self.node("MaxPool", [ps[-1].name], "pool", kernel_shape=[5, 5], strides=[1, 1], pads=[2, 2, 2, 2]), y.c, y.hw))
        return self.conv(self.cat(ps), cout, 1)

    def psa(self, x):
        """position-sensitive attention over the 20 x 20 map: heads of 64 channels, keys / queries of 32"""
        c, hw = x.c, x.hw
        heads, kd, hd = c // 64, 32, 64
        n = hw * hw
        qkv = self.conv(x, heads * (2 * kd + hd), 1, act=False)
        r = self.node("Reshape", [qkv.name, self.const(np.array([self.batch, heads, 2 * kd + hd, n], np.int64), "shp")], "qkvr")
        q, k, v = self.node("Split", [r, self.const(np.array([kd, kd, hd], np.int64), "sp")], "qkv", nout=3, axis=2)
        qt = self.node("Transpose", [q], "qt", perm=[0, 1, 3, 2])                     # [N, h, n, kd]
        sc = self.node("MatMul", [qt, k], "sc")                                        # [N, h, n, n]
        sc = self.node("Mul", [sc, self.const(np.array([kd ** -0.5], np.float32), "scale")], "scs")
        pr = self.node("Softmax", [sc], "pr", axis=-1)
        prt = self.node("Transpose", [pr], "prt", perm=[0, 1, 3, 2])
        av = self.node("MatMul", [v, prt], "av")                                       # [N, h, hd, n]
        self.macs += heads * n * n * (kd + hd)
        shp = self.const(np.array([self.batch, c, hw, hw], np.int64), "shp")
        av = T(self.node("Reshape", [av, shp], "avr"), c, hw)
        vm = T(self.node("Reshape", [v, shp], "vr"), c, hw)
        pe = self.conv(vm, c, 3, act=False, groups=c)
        y = self.add(x, self.conv(self.add(av, pe), c, 1, act=False))
        return self.add(y, self.conv(self.conv(y, 2 * c, 1), c, 1, act=False))

    def c2psa(self, x):
        a, b = self.split2(self.conv(x, x.c, 1))
        return self.conv(self.cat([a, self.psa(b)]), x.c, 1)

    def up(self, x):
        y = self.node("Resize", [x.name, "", self.const(np.array([1, 1, 2, 2], np.float32), "scales")], "up", mode="nearest",
                      coordinate_transformation_mode="asymmetric", nearest_mode="floor")
        return T(y, x.c, x.hw * 2)

    # ---- the network
    def build(self):
        pb, N = self.pb, self.batch
        x = T("images", 3, self.size)
        x = self.conv(x, 16, 3, 2)
        x = self.conv(x, 32, 3, 2)
        x = self.c3k2(x, 64, False, 0.25)
        x = self.conv(x, 64, 3, 2)
        p3 = self.c3k2(x, 128, False, 0.25)
        x = self.conv(p3, 128, 3, 2)
        p4 = self.c3k2(x, 128, True)
        x = self.conv(p4, 256, 3, 2)
        x = self.c3k2(x, 256, True)
        x = self.sppf(x, 256)
        p5 = self.c2psa(x)
        n4 = self.c3k2(self.cat([self.up(p5), p4]), 128, False)
        n3 = self.c3k2(self.cat([self.up(n4), p3]), 64, False)
        m4 = self.c3k2(self.cat([self.conv(n3, 64, 3, 2), n4]), 128, False)
        m5 = self.c3k2(self.cat([self.conv(m4, 128, 3, 2), p5]), 256, True)
        levels = [n3, m4, m5]
        c2, c3, c4 = 64, max(levels[0].c, min(self.nc, 100)), max(levels[0].c // 4, self.nm)
        rows = []
        for f in levels:
            box = self.conv(self.conv(self.conv(f, c2, 3), c2, 3), 4, 1, act=False)
            cls = self.conv(self.conv(f, f.c, 3, groups=f.c), c3, 1)
            cls = self.conv(self.conv(self.conv(cls, c3, 3, groups=c3), c3, 1), self.nc, 1, act=False)
            msk = self.conv(self.conv(self.conv(f, c4, 3), c4, 3), self.nm, 1, act=False)
            allc = self.cat([box, cls, msk])
            rows.append(self.node("Reshape", [allc.name, self.const(np.array([N, allc.c, f.hw * f.hw], np.int64), "shp")], "lvl"))
        anchors = sum(f.hw * f.hw for f in levels)
        pred = self.node("Concat", rows, "pred", axis=2)                                       # [N, 4 + nc + nm, anchors]
        pred = self.node("Transpose", [pred], "predt", perm=[0, 2, 1])                           # [N, anchors, 4 + nc + nm]
        box, cls, coef = self.node("Split", [pred, self.const(np.array([4, self.nc, self.nm], np.int64), "sp")], "heads", nout=3, axis=2)
        prob = self.node("Sigmoid", [cls], "prob")
        best = self.node("ReduceMax", [prob], "best", axes=[2], keepdims=0)                      # [N, anchors]
        kk = min(300, anchors)
        top, idx = self.node("TopK", [best, self.const(np.array([kk], np.int64), "k")], "top", nout=2, axis=-1, largest=1, sorted=1)
        idx3 = self.node("Unsqueeze", [idx, self.const(np.array([2], np.int64), "ax")], "idx3")   # [N, k, 1]

        def take(src, width, tag):
            ix = self.node("Expand", [idx3, self.const(np.array([N, kk, width], np.int64), "shp")], tag + "_ix")
            return self.node("GatherElements", [src, ix], tag, axis=1)
        gbox, gprob, gcoef = take(box, 4, "gbox"), take(prob, self.nc, "gprob"), take(coef, self.nm, "gcoef")
        _, klass = self.node("TopK", [gprob, self.const(np.array([1], np.int64), "k")], "klass", nout=2, axis=-1, largest=1, sorted=1)
        klass = self.node("Cast", [klass], "klassf", to=pb.FLOAT)
        score = self.node("Unsqueeze", [top, self.const(np.array([2], np.int64), "ax")], "score")
        det = self.node("Concat", [gbox, score, klass, gcoef], "det", axis=2)                    # [N, k, 4 + 1 + 1 + nm]
        # prototypes: 3 x 3, transposed 2 x 2 stride 2, 3 x 3, 1 x 1 on the stride-8 map
        p = self.conv(n3, 64, 3)
        wt = (self.rng.standard_normal((64, 64, 2, 2)) * np.sqrt(1.0 / 64)).astype(np.float32)
        pt = self.node("ConvTranspose", [p.name, self.const(wt, "w"), self.const(np.zeros(64, np.float32), "b")], "protoT", kernel_shape=[2, 2],
                       strides=[2, 2])
        self.macs += 64 * 64 * 4 * p.hw * p.hw
        self.convs += 1
        proto = self.conv(self.conv(T(pt, 64, p.hw * 2), 64, 3), self.nm, 1)
        self.g.node += [pb.Node("Identity", [det], ["detections"]), pb.Node("Identity", [proto.name], ["mask_features"])]
        self.g.output = [pb.ValueInfo("detections", pb.FLOAT, [N, kk, 6 + self.nm]),
                         pb.ValueInfo("mask_features", pb.FLOAT, [N, self.nm, proto.hw, proto.hw])]
        return pb.Model(self.g, opset=17).serialize()


def yolo_onnx(batch, size=640, seed=26):
    b = Builder(batch, size, seed)
    data = b.build()
    return data, {"convolutions": b.convs, "gmacs_per_image": round(b.macs / 1e9, 3), "gflop_per_image": round(2 * b.macs / 1e9, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--check", type=int, default=-1, help="images of the batch compared with the batch-1 plan (default: all)")
    ap.add_argument("--compile-only", action="store_true")
    ap.add_argument("--no-fold", action="store_true", help="keep Concat / Split along C as copy kernels (default: channel views, plan.fold_channel_views)")
    ap.add_argument("--no-batch1", action="store_true", help="skip the batch-1 plan (check and timing): a clean kernel trace of the batch-N graph")
    ap.add_argument("--table", default=None, help="write a per-statement table (device ms by HIP events, GFLOP, MB, the statement's own bound) here")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from lele_amd.compiler import compile_model
    data, info = yolo_onnx(args.batch, args.size)
    plan, blob = compile_model(data, "yolo26n_seg_shaped_n%d" % args.batch)
    fns = {}
    for st in plan["statements"]:
        fns[st.get("fn", st.get("op", "?"))] = fns.get(st.get("fn", st.get("op", "?")), 0) + 1
    rec = {"model": "Yolo26n-seg-shaped (tools/yolo_graph.py), synthetic weights", "batch": args.batch, "input": [args.batch, 3, args.size, args.size],
           "onnx_bytes": len(data), **info, "kernel_calls": len(plan["statements"]), "calls_by_kernel": dict(sorted(fns.items(), key=lambda kv: -kv[1]))}
    if args.compile_only:
        print(json.dumps(rec))
        return
    import lele_amd
    from lele_amd.plan import Runner, load_weights_bin as load_weights
    from lele_amd.tensor import TensorView
    ctx = lele_amd.default_ctx(0)
    rng = np.random.default_rng(64)
    images = rng.uniform(0, 1, (args.batch, 3, args.size, args.size)).astype(np.float32)
    big = Runner(plan, load_weights(plan, blob), ctx)
    xb = ctx.buf().upload(images)
    feed = {"images": TensorView(xb)}
    big.shapes = {}
    outs = [o.numpy().copy() for o in big.run(feed)]
    rec["finite"] = bool(all(np.isfinite(o).all() for o in outs))
    rec["outputs"] = [list(o.shape) for o in outs]
    shapes = big.shapes
    if not args.no_fold:   # Concat / Split along C as views of one buffer: the same bits, fewer kernels
        from lele_amd.plan import fold_channel_views
        folded = fold_channel_views(plan, shapes)
        fr = Runner(folded, load_weights(folded, blob), ctx)
        same = all(np.array_equal(a, o.numpy()) for a, o in zip(outs, fr.run(feed)))
        rec.update({"channel_views": folded["folded"], "folded_equals_unfolded_bitwise": bool(same), "kernel_calls_folded": fr.calls})
        if same:
            for b_ in big.ws.values():
                b_.close()
            big = fr
    if args.table:
        big.stmt_times = []
        big.run(feed)
        big.stmt_times = []
        big.run(feed)
        stmt_times, big.stmt_times = big.stmt_times, None   # off again: the stopwatch cannot run inside a graph capture
        rows = []
        byname = {}
        for st in big.plan["statements"]:
            for o in st.get("out", []):
                byname[o] = st
        for idx, fn, o, ms in stmt_times:
            st = byname[o]
            osh = shapes.get(o, [])
            nbytes = 4 * int(np.prod(osh)) if osh else 0
            gflop = 0.0
            geo = ""
            for a in st.get("args", []):
                if isinstance(a, dict) and "ref" in a and a["ref"] in shapes:
                    nbytes += 4 * int(np.prod(shapes[a["ref"]]))
                elif isinstance(a, dict) and "list" in a:
                    nbytes += sum(4 * int(np.prod(shapes[v["ref"]])) for v in a["list"] if isinstance(v, dict) and v.get("ref") in shapes)
            if fn.startswith("conv") and len(osh) == 4:
                w = st["args"][1]["weight"][3]
                xs = shapes[st["args"][0]["ref"]]
                if fn == "conv_transpose":
                    gflop = 2.0 * xs[0] * xs[2] * xs[3] * w[0] * w[1] * w[2] * w[3] / 1e9
                else:
                    gflop = 2.0 * int(np.prod(osh)) * w[1] * w[2] * w[3] / 1e9
                sh_ = 1 if fn == "conv2d_res" else 0   # the residual sits in front of the attributes
                geo = "%d->%d k%d s%s g%s @%dx%d" % (xs[1], osh[1], w[2], st["args"][6 + sh_]["list"][0]["int"] if len(st["args"]) > 6 + sh_ and st["args"][6 + sh_].get("list") else "?",
                                                  st["args"][4 + sh_].get("int", "?") if len(st["args"]) > 4 + sh_ else "?", osh[2], osh[3])
            t_mfma, t_hbm = gflop / 157.3, nbytes / 6.0e9   # ms at the f32 MFMA peak (157.3 GFLOP per ms) / at 6 TB/s
            rows.append({"stmt": idx, "fn": fn, "out": o, "shape": osh, "geometry": geo, "ms": round(ms, 4), "gflop": round(gflop, 3), "mbytes": round(nbytes / 1e6, 2),
                         "bound_ms": round(max(t_mfma, t_hbm), 4), "bound": "mfma" if t_mfma > t_hbm else "hbm", "frac": round(max(t_mfma, t_hbm) / ms, 3) if ms > 0 else None})
        rows.sort(key=lambda r: -r["ms"])
        tot = sum(r["ms"] for r in rows)
        os.makedirs(os.path.dirname(os.path.abspath(args.table)), exist_ok=True)
        json.dump({"total_ms_eager_events": round(tot, 3), "sum_of_bounds_ms": round(sum(r["bound_ms"] for r in rows), 3), "rows": rows}, open(args.table, "w"), indent=0)
        for r in rows[:40]:
            print("%7.3f ms  %-16s %-28s %8.2f GFLOP %8.1f MB  bound %-4s %6.3f ms  frac %s" % (r["ms"], r["fn"], r["geometry"] or str(r["shape"]), r["gflop"], r["mbytes"],
                                                                                             r["bound"], r["bound_ms"], r["frac"]), file=sys.stderr)
        print("eager total %.3f ms, sum of bounds %.3f ms" % (tot, sum(r["bound_ms"] for r in rows)), file=sys.stderr)
    # the batch-1 plan of the same network (same seed -> same weights), image by image
    d1, _ = yolo_onnx(1, args.size)
    p1, b1 = compile_model(d1, "yolo26n_seg_shaped_n1")
    one = Runner(p1, load_weights(p1, b1), ctx)
    x1 = ctx.buf()
    ncheck = args.batch if args.check < 0 else min(args.check, args.batch)
    if args.no_batch1:
        ncheck = 0
    # the prototype map is a convolution stack: value for value.  The detections pass through two top-k selections: the scores of
    # the 300 selected anchors are compared in order, the rows only where both forwards selected the same anchor (two anchors
    # whose scores differ in the last bits may swap places between the two kernels' summation orders)
    names = [o["name"] if isinstance(o, dict) else str(o) for o in plan.get("outputs", [])] or ["detections", "mask_features"]
    worst, bits, swapped = [0.0] * len(outs), True, 0

    def bars(a, b):
        den = 1e-4 * np.maximum(np.abs(a), float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))) + 1e-7
        return float((np.abs(a - b) / den).max()) if a.size else 0.0
    for i in range(ncheck):
        o1 = [o.numpy() for o in one.run({"images": TensorView(x1.upload(images[i:i + 1]))})]
        for j, (a, b) in enumerate(zip(o1, outs)):
            bi = b[i:i + 1]
            bits = bits and bool(np.array_equal(a, bi))
            if a.ndim == 3 and a.shape[-1] == 38:   # [1, 300, 38]: box 4, score, class, 32 coefficients
                worst[j] = max(worst[j], bars(a[..., 4], bi[..., 4]))
                same = np.abs(a[0, :, :4] - bi[0, :, :4]).max(axis=1) <= 1e-3 * (1 + np.abs(a[0, :, :4]).max(axis=1))
                swapped += int((~same).sum())
                worst[j] = max(worst[j], bars(a[0][same], bi[0][same]))
            else:
                worst[j] = max(worst[j], bars(a, bi))
    rec.update({"images_checked_against_the_batch_1_plan": ncheck, "max_error_in_units_of_1e-4_per_output": [round(w, 4) for w in worst],
                "detection_rows_in_a_different_order": swapped, "bit_identical": bits})
    # timing: the batch-N forward as one recorded graph
    ctx.sync()
    ctx.graph_begin()
    big.run(feed)
    gr = ctx.graph_end()
    gr.launch()
    ctx.sync()
    ctx.timer_start()
    for _ in range(args.runs):
        gr.launch()
    ms = ctx.timer_stop() / args.runs
    gr.close()
    flop = 2 * info["gmacs_per_image"] * 1e9 * args.batch
    rec.update({"graph_ms_per_forward": round(ms, 3), "images_per_s": round(args.batch / ms * 1e3, 1), "tflops_f32": round(flop / ms / 1e9, 2),
                "f32_mfma_peak_tflops": 157.3, "fraction_of_the_f32_mfma_peak": round(flop / ms / 1e9 / 157.3, 3),
                "floor_ms_at_the_f32_mfma_peak": round(flop / 157.3e12 * 1e3, 3)})
    if not args.no_batch1:
        feed1 = {"images": TensorView(x1.upload(images[:1]))}
        ctx.sync()
        ctx.graph_begin()
        one.run(feed1)
        g1 = ctx.graph_end()
        g1.launch()
        ctx.sync()
        ctx.timer_start()
        for _ in range(args.runs):
            g1.launch()
        ms1 = ctx.timer_stop() / args.runs
        g1.close()
        rec.update({"batch_1_graph_ms_per_forward": round(ms1, 3), "batch_1_images_per_s": round(1e3 / ms1, 1)})
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
