#!/usr/bin/env python3
"""Independent sub-batch chains as lanes: a configs[3] shard (32 x 10 s) as 1, 2 and 4 INDEPENDENT groups of
utterances, each a linear 70-layer chain, recorded as parallel branches of ONE hipGraph (one fork, one join per forward -- not a pair
per layer as tools/dag_bench.py does); the same for configs[2] with 1, 2, 4 utterances of 30 s in flight.

Every quantiser of the model works per utterance (src/kernels/quantization.rs:104-128: the range is taken per batch slice), so an
utterance's logits do not depend on which group it rides in: the groups' logits are compared bit for bit with the single chain's.

    python tools/chains_bench.py --out gpurun_out/chains_bench.json"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def record_chains(ctx, runners, feeds):
    """-> (graph, [logits TensorView per chain]).  Chain 0 runs on lane 0, chain i on lane i; lanes 1.. start after the point lane 0 has
    reached when the forward begins (their inputs were produced on lane 0) and lane 0 ends after all of them."""
    k = len(runners)
    base = ctx.lane_events(k)
    outs = [None] * k

    def forward():
        ctx.lane_record(base)
        for i in range(1, k):
            ctx.lane_set(i)
            ctx.lane_wait(base)
            outs[i] = runners[i].run(feeds[i])[0]
            ctx.lane_record(base + i)
        ctx.lane_set(0)
        outs[0] = runners[0].run(feeds[0])[0]
        for i in range(1, k):
            ctx.lane_wait(base + i)

    forward()                 # eager once with its lanes: packs weights, sizes every lane's staging
    ctx.sync()
    ctx.graph_begin()
    forward()
    g = ctx.graph_end()
    return g, outs, base


def graph_ms(ctx, g, runs):
    for _ in range(3):
        g.launch()
    ctx.sync()
    best = 1e9
    for _ in range(3):
        ctx.timer_start()
        for _ in range(runs):
            g.launch()
        best = min(best, ctx.timer_stop() / runs)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=70)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--single", action="store_true", help="one chain only: the plain graph times of the three workloads")
    args = ap.parse_args()
    import lele_amd
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.tensor import TensorView
    from sensevoice_graph import Encoder, encoder_onnx
    ctx = lele_amd.default_ctx(0)
    enc = Encoder(ctx, args.layers, damped=True)
    recs = []
    plans = {}

    def plan_for(b):
        if b not in plans:
            plan, blob = compile_model(encoder_onnx(enc, b), "sv")
            plans[b] = (plan, load_weights_bin(plan, blob))
        return plans[b]

    for name, total, t, splits in (("configs[3] shard: 32 x 10 s", 32, 171, (1, 2, 4)),
                                   ("configs[2] in flight: 4 x 30 s (utterances of DIFFERENT requests; 1 chain = one batch of 4)", 4, 504, (1, 2, 4)),
                                   ("configs[2] alone: 1 x 30 s", 1, 504, (1,))):
        x = np.random.default_rng(t + total).standard_normal((total, t, 560)).astype(np.float32)
        rec = {"workload": name, "layers": args.layers}
        want = None
        for k in (splits[:1] if args.single else splits):
            per = total // k
            plan, w = plan_for(per)
            runners = [Runner(plan, w, ctx) for _ in range(k)]
            feeds = [{"feats": TensorView(ctx.buf().upload(x[i * per:(i + 1) * per]))} for i in range(k)]
            g, outs, base = record_chains(ctx, runners, feeds)
            ms = graph_ms(ctx, g, args.runs)
            got = np.concatenate([o.numpy() for o in outs], 0)
            if want is None:
                want = got
            rec["chains_%d" % k] = {"utterances_per_chain": per, "graph_ms": round(ms, 4), "ms_per_utterance": round(ms / total, 4),
                                    "bit_identical_to_one_chain": bool(np.array_equal(got, want))}
            g.close()
            ctx.lane_events_release(base, k)
            for r in runners:
                for b_ in r.ws.values():
                    b_.close()
                r.close()
        rec["best"] = min((v["graph_ms"], k_) for k_, v in rec.items() if k_.startswith("chains_"))[1]
        print(json.dumps(rec), flush=True)
        recs.append(rec)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(recs, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
