#!/usr/bin/env python3
"""Linear hipGraph vs DAG hipGraph (lele_amd/lanes.py) of the same plan: configs[2] (one 30 s utterance), a configs[3] shard (32 x 10 s),
configs[4] (look-alike and, when _lifted/ is present, the reference's generated graph) at batch 64.  Per plan: bit-identity of the
outputs, graph replay time (HIP events) both ways.

    python tools/dag_bench.py --out gpurun_out/dag_bench.json [--lanes 3]"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def graph_ms(ctx, runner, feed, runs):
    ctx.sync()
    ctx.graph_begin()
    res = runner.run(feed)
    g = ctx.graph_end()
    for _ in range(3):
        g.launch()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(runs):
            g.launch()
        best = min(best, ctx.timer_stop() / runs)
    out = [o.numpy().copy() for o in res]
    g.close()
    return best, out


def compare(ctx, name, plan, weights, feed, lanes, runs, gain):
    from lele_amd.lanes import schedule
    from lele_amd.plan import Runner
    r0 = Runner(plan, weights, ctx)
    r0.run(feed)
    r0.stmt_times = []
    r0.stmt_repeat = 8
    r0.run(feed)
    times = {o: ms for _i, _fn, o, ms in r0.stmt_times}
    r0.stmt_times, r0.stmt_repeat = None, 1
    lin_ms, want = graph_ms(ctx, r0, feed, runs)
    rec = {"plan": name, "linear_graph_ms": round(lin_ms, 4), "eager_sum_of_statement_ms": round(sum(times.values()), 4)}
    for k in lanes:
        dag = schedule(plan, times, lanes=k, min_gain_ms=gain)
        if dag is None:
            rec["dag"] = "not schedulable"
            break
        r1 = Runner(dag, weights, ctx)
        r1.run(feed)
        ms, got = graph_ms(ctx, r1, feed, runs)
        rec["lanes_%d" % k] = {"graph_ms": round(ms, 4), "speedup": round(lin_ms / ms, 4), "bit_identical": bool(all(np.array_equal(a, b) for a, b in zip(got, want))),
                               **dag["dag"]}
        for b_ in r1.ws.values():
            b_.close()
    print(json.dumps(rec), flush=True)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lanes", default="2,3")
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--gain", type=float, default=0.004)
    ap.add_argument("--layers", type=int, default=70)
    ap.add_argument("--only", default="")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    lanes = [int(v) for v in args.lanes.split(",")]
    import lele_amd
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, fold_channel_views, load_weights_bin
    from lele_amd.tensor import TensorView
    ctx = lele_amd.default_ctx(0)
    recs = []
    want = set(args.only.split(",")) if args.only else None
    if want is None or "sv" in want:
        from sensevoice_graph import Encoder, encoder_onnx
        enc = Encoder(ctx, args.layers)
        for name, b, t in (("configs[2]: 1 x 30 s", 1, 504), ("configs[3] shard: 32 x 10 s", 32, 171)):
            plan, blob = compile_model(encoder_onnx(enc, b), "sv")
            x = np.random.default_rng(t).standard_normal((b, t, 560)).astype(np.float32)
            recs.append(compare(ctx, name, plan, load_weights_bin(plan, blob), {"feats": TensorView(ctx.buf().upload(x))}, lanes, args.runs, args.gain))
    if want is None or "yolo" in want:
        from yolo_graph import yolo_onnx
        nb = 64
        plan, blob = compile_model(yolo_onnx(nb)[0], "yolo")
        w = load_weights_bin(plan, blob)
        images = np.random.default_rng(5).uniform(0, 1, (nb, 3, 640, 640)).astype(np.float32)
        feed = {"images": TensorView(ctx.buf().upload(images))}
        r0 = Runner(plan, w, ctx)
        r0.shapes = {}
        r0.run(feed)
        folded = fold_channel_views(plan, r0.shapes)
        for b_ in r0.ws.values():
            b_.close()
        recs.append(compare(ctx, "configs[4] look-alike, batch 64, channel views", folded, w, feed, lanes, args.runs, args.gain))
    lifted = os.path.join(ROOT, "_lifted", "yolo26seg_plan.json")
    if (want is None or "lifted" in want) and os.path.exists(lifted):
        import yolo_lifted_batch as Y
        one, big, lfeed, limages, lname, louts, lrec = Y.build(ctx, lifted, 64)
        recs.append(compare(ctx, "configs[4] reference graph, batch 64, channel views", big.plan, big.raw, lfeed, lanes, args.runs, args.gain))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(recs, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
