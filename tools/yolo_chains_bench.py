#!/usr/bin/env python3
"""configs[4] as independent sub-batch chains: the reference's generated Yolo26n-seg graph at batch 64 as 1, 2 and 4 groups of images,
each a linear chain of the folded plan, recorded as parallel branches of ONE hipGraph (one fork, one join per forward).  An image's
outputs do not depend on which group it rides in, so the groups' outputs are compared bit for bit with the single chain's.

What it is for: a forward of b images costs ~2.0 ms + 0.098 ms x b (profiles/r06_yolo_batch_sweep.json) -- a quarter of the batch-64
forward is per-kernel fixed cost (ramp, tail, dependent-launch gap of 192 kernels).  Two chains side by side can fill each other's ramps and tails.

    python tools/yolo_chains_bench.py --out gpurun_out/yolo_chains.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", default=os.path.join(ROOT, "_lifted", "yolo26seg_plan.json"))
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--splits", default="1,2,4")
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import lele_amd
    from lele_amd.plan import Runner
    from lele_amd.tensor import TensorView
    from chains_bench import graph_ms, record_chains
    from yolo_lifted_batch import build
    ctx = lele_amd.default_ctx(0)
    rng = np.random.default_rng(64)
    images = rng.uniform(0, 1, (args.batch, 3, 640, 640)).astype(np.float32)
    rec = {"workload": "lele-generated Yolo26n-seg, batch %d, folded linear plan per chain" % args.batch}
    want = None
    for k in [int(s) for s in args.splits.split(",")]:
        per = args.batch // k
        one, big, feed, _, name, _, _ = build(ctx, args.plan, per)
        runners = [big] + [Runner(big.plan, big.raw, ctx) for _ in range(k - 1)]
        feeds = [{name: TensorView(ctx.buf().upload(images[i * per:(i + 1) * per]))} for i in range(k)]
        base = ctx.lane_events(k)
        outs = [None] * k

        def forward():
            ctx.lane_record(base)
            for i in range(1, k):
                ctx.lane_set(i)
                ctx.lane_wait(base)
                outs[i] = runners[i].run(feeds[i])
                ctx.lane_record(base + i)
            ctx.lane_set(0)
            outs[0] = runners[0].run(feeds[0])
            for i in range(1, k):
                ctx.lane_wait(base + i)

        forward()
        ctx.sync()
        ctx.graph_begin()
        forward()
        g = ctx.graph_end()
        ms = graph_ms(ctx, g, args.runs)
        got = [np.concatenate([o[j].numpy() for o in outs], 0) for j in range(len(outs[0]))]
        if want is None:
            want = got
        rec["chains_%d" % k] = {"images_per_chain": per, "graph_ms": round(ms, 4),
                                "bit_identical_to_one_chain": bool(all(np.array_equal(a, b) for a, b in zip(got, want)))}
        print(json.dumps(rec["chains_%d" % k]), flush=True)
        g.close()
        ctx.lane_events_release(base, k)
        for r in runners + [one]:
            for b_ in r.ws.values():
                b_.close()
            r.close()
    print(json.dumps(rec))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
