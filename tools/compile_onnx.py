#!/usr/bin/env python3
"""Compile an ONNX model for the device and (optionally) run it: lele's `cargo run --bin lele_gen` + generated crate, as a
plan.

    compile: python tools/compile_onnx.py compile model.onnx -o out_dir      -> out_dir/<name>_plan.json, <name>_weights.bin
    run    : python tools/compile_onnx.py run out_dir/<name>_plan.json [--shape x=1,3,640,640] [--runs 10]     (needs a GPU)

The weights file has lele's layout (src/compiler/mod.rs:1381-1505), so a plan lifted from lele-generated Rust
(tools/lift_generated.py) can run on the weights.bin written here and vice versa.  Inputs for `run` are synthetic
(uniform [0, 1) for f32, zeros for integer inputs); symbolic dimensions must be given with --shape.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("compile")
    a.add_argument("model")
    a.add_argument("-o", "--out", required=True)
    a.add_argument("--name", default=None)
    a.add_argument("--bind", action="append", default=[], help="input=v[,v...]: fix an integer graph input at compile time, e.g. sr=16000 "
                   "(an `If` on it then inlines the taken branch)")
    b = sub.add_parser("run")
    b.add_argument("plan")
    b.add_argument("--weights", default=None, help="default: the _weights.bin next to the plan")
    b.add_argument("--shape", action="append", default=[], help="input=dims, e.g. x=1,3,640,640 (required for symbolic dims)")
    b.add_argument("--runs", type=int, default=10)
    args = ap.parse_args()
    if args.cmd == "compile":
        from lele_amd.compiler import compile_model
        name = args.name or os.path.splitext(os.path.basename(args.model))[0]
        t0 = time.perf_counter()
        bind = {kv.split("=")[0]: np.array([int(v) for v in kv.split("=")[1].split(",")], np.int64) for kv in args.bind}
        plan, blob = compile_model(args.model, name, bind=bind or None)
        os.makedirs(args.out, exist_ok=True)
        json.dump(plan, open(os.path.join(args.out, name + "_plan.json"), "w"))
        open(os.path.join(args.out, name + "_weights.bin"), "wb").write(blob)
        calls = {}
        for st in plan["statements"]:
            key = st.get("fn", "host:" + st.get("onnx", "")) if st["op"] in ("call", "host") else st["op"]
            calls[key] = calls.get(key, 0) + 1
        print(json.dumps({"model": name, "statements": len(plan["statements"]), "slots": len(plan["slots"]),
                          "weights_bin_bytes": len(blob), "compile_s": round(time.perf_counter() - t0, 2), "calls": calls}))
        return
    plan = json.load(open(args.plan))
    wpath = args.weights or args.plan.replace("_plan.json", "_weights.bin")
    import lele_amd
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.tensor import TensorView
    ctx = lele_amd._lib.Ctx(0)
    r = Runner(plan, load_weights_bin(plan, wpath), ctx)
    shapes = dict(kv.split("=") for kv in args.shape)
    rng = np.random.default_rng(0)
    inputs = {}
    for info in plan.get("input_info", []):
        dims = [int(v) for v in shapes[info["name"]].split(",")] if info["name"] in shapes else info["shape"]
        if dims is None or any(not isinstance(d, int) for d in dims):
            raise SystemExit("input %s has symbolic shape %s: pass --shape %s=..." % (info["name"], dims, info["name"]))
        if info["dtype"] == "i64":
            inputs[info["name"]] = np.zeros(dims, np.int64)
        else:
            inputs[info["name"]] = TensorView(ctx.buf().upload(rng.uniform(0, 1, dims).astype(np.float32)))
    outs = r.run(inputs)
    rec = {"model": plan["source"], "outputs": {o: list(v.shape) for o, v in zip(plan["outputs"], outs)},
           "finite": all(bool(np.isfinite(np.asarray(v.numpy(), np.float64)).all()) for v in outs), "kernel_calls": r.calls}
    r.run(inputs)
    ts = []
    for _ in range(args.runs):
        ctx.sync()
        t0 = time.perf_counter()
        r.run(inputs)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    rec["eager_ms"] = round(1e3 * float(np.mean(ts)), 3)
    try:
        ctx.sync()
        ctx.graph_begin()
        r.run(inputs)
        g = ctx.graph_end()
        g.launch()
        ts = []
        for _ in range(args.runs):
            ctx.sync()
            t0 = time.perf_counter()
            g.launch()
            ctx.sync()
            ts.append(time.perf_counter() - t0)
        rec["graph_ms"] = round(1e3 * float(np.mean(ts)), 3)
    except Exception as e:  # noqa: BLE001  (a plan with run-time host tensors feeding device ops cannot be captured)
        ctx.graph_abort()
        rec["graph_ms"] = "capture failed: %s" % e
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
