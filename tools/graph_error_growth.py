#!/usr/bin/env python3
"""Where a graph-level difference between the device and the CPU oracle comes from: the reference's generated Yolo26n-seg call
sequence (lifted plan, calibrated synthetic weights) run at batch N on the device and, image 0, on the oracle (oracle/plan_ref.py);
after EVERY statement the device value of image 0 is compared with the oracle's value of the same name.  Two columns:

  chained : device statement fed by the device's own previous results (what a forward does) -- the error as it grows through depth;
  local   : the same device kernel fed the ORACLE's inputs for that statement (uploaded) -- the error the statement itself adds.

    python tools/graph_error_growth.py --batch 64 --out gpurun_out/graph_error_growth.json      (test infrastructure: uses oracle/)"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def bars(got, want, tol=1e-4):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    if got.shape != want.shape or not want.size:
        return None
    rms = float(np.sqrt(np.mean(np.square(want))))
    d = np.abs(got - want)
    b = d / (tol * np.abs(want) + tol * rms + 1e-7)
    return {"max_bars": round(float(b.max()), 3), "over_1": int((b > 1).sum()), "max_abs": float(d.max()), "rms": rms}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", default=os.path.join(ROOT, "_lifted", "yolo26seg_plan.json"))
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--out", default=None)
    ap.add_argument("--local", action="store_true", help="also run every convolution on the oracle's inputs (the statement's own error)")
    args = ap.parse_args()
    import lele_amd
    import lift_generated as L
    from lele_amd import kernels as K
    from lele_amd.plan import Runner, fuse_sigmoid_mul, rebatch_lifted, replan_lifted
    from lele_amd.tensor import TensorView
    from oracle import plan_ref
    ctx = lele_amd.default_ctx(0)
    n = args.batch
    plan = json.load(open(args.plan))
    name = plan["inputs"][-1]
    rng = np.random.default_rng(64)
    images = rng.uniform(0, 1, (n, 3, 640, 640)).astype(np.float32)
    raw = plan_ref.calibrate(plan, L.synth_weights(plan, dict(L.DEFAULT_CONSTS)), {name: images[:1]})
    r1 = Runner(plan, raw, ctx)
    r1.shapes = {}
    r1.run({name: TensorView(ctx.buf().upload(images[:1]))})
    p2 = replan_lifted(fuse_sigmoid_mul(plan, r1.shapes), r1.shapes)
    w2 = {k: raw[int(k.split(":")[0])] for k in p2["weights"]}
    pn = rebatch_lifted(p2, n, r1.shapes) if n > 1 else p2
    # the oracle runs the RE-PLANNED batch-1 plan (same statement names as the device's batch-N plan)
    ref = plan_ref.PlanRef(p2, w2)
    ref.taps = {o: None for st in p2["statements"] for o in st.get("out", [])}
    ref.run({name: images[:1]})
    want = ref.taps
    big = Runner(pn, w2, ctx)
    rows = []

    class Watch(dict):   # Runner.taps: `taps[name] = host copy` after the statement that writes `name` -- compare on arrival, keep nothing
        def __setitem__(self, k, got):
            if got is not None and want.get(k) is not None:
                b = bars(got[:1] if got.shape[:1] == (n,) and want[k].shape[:1] == (1,) else got, want[k])
                if b:
                    rows.append(dict(b, out=k, shape=list(want[k].shape)))
            dict.__setitem__(self, k, None)
    big.taps = Watch()
    for k in want:
        dict.__setitem__(big.taps, k, None)
    feed = {name: TensorView(ctx.buf().upload(images))}
    big.run(feed)
    fn_of = {o: st.get("fn", st["op"]) for st in pn["statements"] for o in st.get("out", [])}
    for r in rows:
        r["fn"] = fn_of.get(r["out"], "?")
    if args.local:
        env = {k: v for k, v in want.items() if v is not None}
        env[name] = images[:1]
        for st in p2["statements"]:
            if st["op"] != "call" or not st["fn"].startswith("conv"):
                continue
            pos = [ref.val(a, env) for a in st["args"] if not ("slot" in a or "buf" in a)]
            dev = getattr(K, st["fn"])(*[(np.repeat(p, n, axis=0) if i == 0 else p) for i, p in enumerate(pos)], ctx=ctx).numpy()
            b = bars(dev[:1], want[st["out"][0]])
            for r in rows:
                if r["out"] == st["out"][0]:
                    r["local_max_bars"] = b["max_bars"]
                    r["local_max_abs"] = b["max_abs"]
    for r in rows:
        print("%-16s %-70s %-22s chained %8.3f bars (%6d over, abs %.2e, rms %.2e)%s" % (
            r["fn"], r["out"][-70:], r["shape"], r["max_bars"], r["over_1"], r["max_abs"], r["rms"],
            ("   local %.3f bars" % r["local_max_bars"]) if "local_max_bars" in r else ""))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump({"batch": n, "env": {k: v for k, v in os.environ.items() if k.startswith("LELE_HIP")}, "rows": rows}, open(args.out, "w"), indent=0)


if __name__ == "__main__":
    main()
