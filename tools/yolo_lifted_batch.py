#!/usr/bin/env python3
"""BASELINE configs[4] on the reference's OWN generated Yolo26n-seg call sequence at batch N, as ONE graph.

The plan is lifted from examples/yolo26n-seg/src/yolo26seg.rs where the reference is mounted (tools/lift_generated.py lift ->
_lifted/yolo26seg_plan.json; an untracked artifact that travels with the working tree).  The generated source bakes batch 1 into its
reshapes; plan.rebatch_lifted rewrites those (and the one gather whose exported form flattens the batch away), plan.replan_lifted
re-assigns the buffers, plan.fold_channel_views turns Concat / Split along C into views.  Every checked image of the batch-N forward
is compared with the batch-1 plan's forward of that image (same synthetic weights).

    python tools/yolo_lifted_batch.py --batch 64 --check 4 --out gpurun_out/yolo26seg_n64.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build(ctx, path, batch, seed=64):
    """-> (batch-1 runner, batch-N runner (folded), record)"""
    import lift_generated as L
    from lele_amd.plan import Runner, fold_channel_views, fuse_sigmoid_mul, rebatch_lifted, replan_lifted
    from lele_amd.tensor import TensorView
    plan = json.load(open(path))
    raw = L.synth_weights(plan, dict(L.DEFAULT_CONSTS))
    name = plan["inputs"][-1]
    rng = np.random.default_rng(seed)
    images = rng.uniform(0, 1, (batch, 3, 640, 640)).astype(np.float32)
    r1 = Runner(plan, raw, ctx)
    r1.shapes = {}
    r1.run({name: TensorView(ctx.buf().upload(images[:1]))})
    shapes1 = r1.shapes
    p2 = replan_lifted(fuse_sigmoid_mul(plan, shapes1), shapes1)
    w2 = {k: raw[int(k.split(":")[0])] for k in p2["weights"]}
    one = Runner(p2, w2, ctx)
    pn = rebatch_lifted(p2, batch, shapes1)
    big = Runner(pn, w2, ctx)
    big.shapes = {}
    xb = ctx.buf().upload(images)
    feed = {name: TensorView(xb)}
    outs = [o.numpy().copy() for o in big.run(feed)]
    rec = {"model": "lele-generated Yolo26n-seg (examples/yolo26n-seg/src/yolo26seg.rs, lifted), synthetic weights", "batch": batch,
           "kernel_calls_batch_1_replanned": one.calls if one.calls else None, "kernel_calls_rebatched": big.calls,
           "outputs": [list(o.shape) for o in outs], "finite": bool(all(np.isfinite(o).all() for o in outs))}
    folded = fold_channel_views(pn, big.shapes)
    fr = Runner(folded, w2, ctx)
    same = all(np.array_equal(a, o.numpy()) for a, o in zip(outs, fr.run(feed)))
    rec.update({"channel_views": folded["folded"], "folded_equals_unfolded_bitwise": bool(same), "kernel_calls_folded": fr.calls})
    if same:
        for b_ in big.ws.values():
            b_.close()
        big = fr
    return one, big, feed, images, name, outs, rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", default=os.path.join(ROOT, "_lifted", "yolo26seg_plan.json"))
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--check", type=int, default=4)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--table", default=None, help="per-statement device times (HIP events) of one eager forward, slowest first")
    args = ap.parse_args()
    if not os.path.exists(args.plan):
        raise SystemExit("no lifted plan at %s (python tools/lift_generated.py lift <reference>/examples/yolo26n-seg/src/yolo26seg.rs -o %s)" % (args.plan, args.plan))
    import lele_amd
    from lele_amd.tensor import TensorView
    ctx = lele_amd.default_ctx(0)
    one, big, feed, images, name, outs, rec = build(ctx, args.plan, args.batch)

    def bars(a, b):
        den = 1e-4 * np.maximum(np.abs(a), float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))) + 1e-7
        return float((np.abs(a - b) / den).max()) if a.size else 0.0
    x1 = ctx.buf()
    worst, bits, swapped = [0.0] * len(outs), True, 0
    idx = sorted(set(np.linspace(0, args.batch - 1, min(args.check, args.batch)).astype(int).tolist()))
    for i in idx:
        o1 = [o.numpy() for o in one.run({name: TensorView(x1.upload(images[i:i + 1]))})]
        for j, (a, b) in enumerate(zip(o1, outs)):
            bi = b[i:i + 1]
            bits = bits and bool(np.array_equal(a, bi))
            if a.ndim == 3 and a.shape[-1] == 38:   # [1, 300, 38]: box 4, score, class, 32 coefficients (two top-k selections upstream)
                worst[j] = max(worst[j], bars(a[..., 4], bi[..., 4]))
                same = np.abs(a[0, :, :4] - bi[0, :, :4]).max(axis=1) <= 1e-3 * (1 + np.abs(a[0, :, :4]).max(axis=1))
                swapped += int((~same).sum())
                worst[j] = max(worst[j], bars(a[0][same], bi[0][same]))
            else:
                worst[j] = max(worst[j], bars(a, bi))
    rec.update({"images_checked_against_the_batch_1_plan": idx, "max_error_in_units_of_1e-4_per_output": [round(w, 4) for w in worst],
                "detection_rows_in_a_different_order": swapped, "bit_identical": bits})
    if args.table:
        big.shapes = {}
        big.run(feed)
        shapes = big.shapes
        big.stmt_times, big.stmt_repeat = [], 8     # trains of 8 launches: a single eager launch of a 10 us kernel reads as 30 us
        big.run(feed)
        times, big.stmt_times, big.stmt_repeat = big.stmt_times, None, 1
        byname = {}
        for st in big.plan["statements"]:
            for o in st.get("out", []):
                byname[o] = st
        rows = []
        for idx, fn, o, ms in times:
            st = byname[o]
            osh = shapes.get(o, [])
            nbytes = 4 * int(np.prod(osh)) if osh else 0
            gflop, geo = 0.0, ""
            for a in st.get("args", []):
                if isinstance(a, dict) and "ref" in a and a["ref"] in shapes:
                    nbytes += 4 * int(np.prod(shapes[a["ref"]]))
            if fn.startswith("conv") and len(osh) == 4 and isinstance(st["args"][1], dict) and "weight" in st["args"][1]:
                w = st["args"][1]["weight"][3]
                xs = shapes.get(st["args"][0].get("ref"), [0, 0, 0, 0])
                if fn == "conv_transpose":
                    gflop = 2.0 * xs[0] * xs[2] * xs[3] * w[0] * w[1] * w[2] * w[3] / 1e9
                else:
                    gflop = 2.0 * int(np.prod(osh)) * w[1] * w[2] * w[3] / 1e9
                sh_ = 1 if fn == "conv2d_res" else 0
                stride = st["args"][6 + sh_]["list"][0]["int"] if len(st["args"]) > 6 + sh_ and st["args"][6 + sh_].get("list") else "?"
                group = st["args"][4 + sh_].get("int", "?") if len(st["args"]) > 4 + sh_ else "?"
                geo = "%d->%d k%d s%s g%s @%dx%d" % (xs[1] if len(xs) > 1 else 0, osh[1], w[2], stride, group, osh[2], osh[3])
            t_mfma, t_hbm = gflop / 157.3, nbytes / 6.0e9
            rows.append({"stmt": idx, "fn": fn, "out": o, "shape": osh, "geometry": geo, "ms": round(ms, 4), "gflop": round(gflop, 3), "mbytes": round(nbytes / 1e6, 2),
                         "bound_ms": round(max(t_mfma, t_hbm), 4), "bound": "mfma" if t_mfma > t_hbm else "hbm", "frac": round(max(t_mfma, t_hbm) / ms, 3) if ms > 0 else None})
        agg = {}
        for r in rows:
            a = agg.setdefault(r["fn"], [0, 0.0, 0.0])
            a[0] += 1
            a[1] += r["ms"]
            a[2] += r["bound_ms"]
        rows.sort(key=lambda r: -(r["ms"] - r["bound_ms"]))
        json.dump({"what": "per-statement device time (HIP events, trains of 8 launches) of the reference's generated Yolo26n-seg graph at batch %d against "
                           "max(f32 MFMA at 157.3 TFLOP/s, HBM at 6 TB/s) of the statement's own geometry; rows by absolute gap" % args.batch,
                   "total_ms": round(sum(r["ms"] for r in rows), 3), "sum_of_bounds_ms": round(sum(r["bound_ms"] for r in rows), 3),
                   "by_function": sorted(([fn, c, round(t, 4), round(b, 4)] for fn, (c, t, b) in agg.items()), key=lambda r: -r[2]), "rows": rows},
                  open(args.table, "w"), indent=0)
        for r in rows[:30]:
            print("%7.3f ms  %-16s %-30s %8.2f GFLOP %8.1f MB  bound %-4s %6.3f ms  frac %s" % (r["ms"], r["fn"], r["geometry"] or str(r["shape"]), r["gflop"], r["mbytes"],
                                                                                             r["bound"], r["bound_ms"], r["frac"]), file=sys.stderr)
    ctx.sync()
    ctx.graph_begin()
    big.run(feed)
    gr = ctx.graph_end()
    for _ in range(3):
        gr.launch()
    ctx.sync()
    ctx.timer_start()
    for _ in range(args.runs):
        gr.launch()
    ms = ctx.timer_stop() / args.runs
    gr.close()
    gflop = 9.127   # the generated graph's 118 convolutions, per image
    rec.update({"graph_ms_per_forward": round(ms, 3), "images_per_s": round(args.batch / ms * 1e3, 1), "gflop_per_image": gflop,
                "tflops_f32": round(gflop * args.batch / ms, 2), "fraction_of_the_f32_mfma_peak": round(gflop * args.batch / ms / 157.3, 3)})
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
