#!/usr/bin/env python3
"""Lift the call sequence out of a lele-GENERATED model source (the `forward` body lele's compiler emits, e.g.
examples/yolo26n-seg/src/yolo26seg.rs) and run it through this library's operator mirror.

lele compiles an ONNX graph to Rust whose body is one `lele::kernels::<op>(...)` statement per node.  Because
`lele_amd.kernels` keeps lele's function names and argument order, the statements can be executed as they stand: this
tool parses each `let X = lele::kernels::fn(args);` into (outputs, fn, argument tree) and interprets it -- nothing about
the model is written by hand.  It is a drop-in check (every op, attribute form and view helper the generated code uses
must exist with the same meaning) and it gives whole-model timings.

    lift:  python tools/lift_generated.py lift /path/to/generated.rs -o _lifted/model_plan.json      (needs the source)
    run :  python tools/lift_generated.py run _lifted/model_plan.json --batch-runs 64                 (needs a GPU)

The plan is a DERIVED, UNTRACKED artifact (like oracle/_ref/): `_lifted/` is git-ignored and never committed; it
exists so that a plan lifted where the reference tree is mounted can be timed on the GPU box, where it is not.
Weights: lele's `<model>_weights.bin` is downloaded at build time and is not in the tree, so the run uses synthetic
weights of the recorded shapes (SURVEY.md section 8(d) recipe); the few integer constants the graph reads from the
weights file (top-k size, resize target, class count) are taken from `--const offset=value,...` or the defaults below.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


# ----------------------------------------------------------------------------------------------- parsing
class P:
    """tiny recursive-descent parser for the argument expressions lele's emitters produce"""

    def __init__(self, s):
        self.s, self.i = s, 0

    def ws(self):
        while self.i < len(self.s) and self.s[self.i].isspace():
            self.i += 1

    def eat(self, tok):
        self.ws()
        if self.s.startswith(tok, self.i):
            self.i += len(tok)
            return True
        return False

    def expect(self, tok):
        if not self.eat(tok):
            raise ValueError("expected %r at ...%s" % (tok, self.s[self.i:self.i + 60]))

    def ident(self):
        self.ws()
        m = re.match(r"[A-Za-z_][A-Za-z_0-9]*", self.s[self.i:])
        if not m:
            raise ValueError("identifier expected at ...%s" % self.s[self.i:self.i + 60])
        self.i += m.end()
        return m.group(0)

    def number(self):
        self.ws()
        m = re.match(r"-?\d+(\.\d+)?([eE][-+]?\d+)?(_?f32|_?usize|_?i64)?", self.s[self.i:])
        if not m:
            raise ValueError("number expected at ...%s" % self.s[self.i:self.i + 60])
        self.i += m.end()
        t = re.sub(r"_?(f32|usize|i64)$", "", m.group(0))
        return {"float": float(t)} if ("." in t or "e" in t.lower()) else {"int": int(t)}

    def int_list(self):  # after '['
        vals = []
        while not self.eat("]"):
            vals.append(self.expr())
            self.eat(",")
        return vals

    def weight(self):  # after 'self.'
        kind = self.ident()
        self.expect("(")
        off = self.number()["int"]
        self.expect(",")
        ln = self.number()["int"]
        self.expect(",")
        self.expect("&")
        self.expect("[")
        shape = [v["int"] for v in self.int_list()]
        self.expect(")")
        node = {"weight": [kind, off, ln, shape]}
        # suffixes:  .data   |   .data[0] as usize   |   .data.iter().map(|&v| v as i64).collect::<Vec<_>>()
        if self.eat(".data"):
            if self.eat("[0]"):
                self.eat(" as usize")
                self.eat("as usize")
                return {"weight_scalar": node["weight"]}
            if self.eat(".iter().map(|&v| v as i64).collect::<Vec<_>>()"):
                return {"weight_list": node["weight"]}
            return {"weight_list": node["weight"]}
        return node

    def expr(self):
        self.ws()
        if self.eat("Some("):
            e = self.expr()
            self.expect(")")
            return {"some": e}
        if self.eat("None"):
            return {"none": True}
        if self.eat("true"):
            return {"bool": True}
        if self.eat("false"):
            return {"bool": False}
        if self.eat('"'):
            j = self.s.index('"', self.i)
            v = self.s[self.i:j]
            self.i = j + 1
            return {"str": v}
        if self.eat("&mut ws."):
            return {"slot": self.ident()}
        if self.eat("&mut "):
            return {"buf": self.ident()}
        if self.eat("&["):
            vals = self.int_list()
            if vals and all("ref" in v for v in vals):
                return {"refs": [v["ref"] for v in vals]}
            return {"list": vals}
        if self.eat("&self."):
            return self.weight()
        if self.eat("self."):
            return self.weight()
        if self.eat("&"):
            return {"ref": self.ident()}
        self.ws()
        if re.match(r"-?\d", self.s[self.i:]):
            return self.number()
        return {"ref": self.ident()}  # bare identifier (e.g. splits_slice)

    def args(self):
        out = []
        self.ws()
        while not self.eat(")"):
            out.append(self.expr())
            self.eat(",")
        return out


def lift(path):
    src = open(path).read()
    m = re.search(r"fn run_chunk_0.*?\{\n(.*?)\n    \}\n", src, re.S)
    if not m:
        raise SystemExit("no run_chunk_0 body found in %s" % path)
    body = m.group(1)
    inputs = re.findall(r"(\w+): TensorView", re.search(r"fn run_chunk_0<[^>]*>\(([^)]*)\)", src).group(1))
    stmts, outputs = [], []
    for line in body.split("\n"):
        line = line.split("//")[0].strip()
        if not line:
            continue
        mm = re.match(r"let (?:mut )?(\w+) = &\[([-\d, ]*)\];$", line)
        if mm:
            stmts.append({"op": "ints", "out": [mm.group(1)], "value": [int(v) for v in mm.group(2).split(",") if v.strip()]})
            continue
        mm = re.match(r"let mut (\w+) = Vec::<f32>::new\(\);$", line)
        if mm:
            stmts.append({"op": "newbuf", "out": [mm.group(1)]})
            continue
        mm = re.match(r"let (\w+) = (\w+)\.swap_remove\((\d+)\);$", line)
        if mm:
            stmts.append({"op": "swap_remove", "out": [mm.group(1)], "list": mm.group(2), "index": int(mm.group(3))})
            continue
        mm = re.match(r"let (\w+) = (\w+)\.clone\(\);$", line)
        if mm:
            stmts.append({"op": "alias", "out": [mm.group(1)], "src": mm.group(2)})
            continue
        mm = re.match(r"let (?:mut )?(\(?[\w, ]+\)?) = (?:lele::kernels::|self\.)(\w+)\((.*);$", line)
        if mm:
            outs = [o.strip() for o in mm.group(1).strip("()").split(",")]
            p = P(mm.group(3))
            stmts.append({"op": "call", "out": outs, "fn": mm.group(2), "args": p.args()})
            continue
        mm = re.match(r"\(([\w.() ,]+)\)$", line)
        if mm:
            outputs = [o.strip().replace(".to_owned()", "") for o in mm.group(1).split(",")]
            continue
        raise SystemExit("cannot parse statement: %s" % line[:200])
    slots = sorted(set(re.findall(r"pub (buf_\d+): Vec<f32>", src)), key=lambda s: int(s.split("_")[1]))
    weights = {}
    def walk(n):
        if isinstance(n, dict):
            for k in ("weight", "weight_scalar", "weight_list"):
                if k in n:
                    kind, off, ln, shape = n[k]
                    weights[off] = [kind, ln, shape]
            for v in n.values():
                walk(v)
        elif isinstance(n, list):
            for v in n:
                walk(v)
    walk(stmts)
    return {"source": os.path.basename(path), "inputs": inputs, "outputs": outputs, "slots": slots, "statements": stmts,
            "weights": {str(k): v for k, v in sorted(weights.items())}}


# ----------------------------------------------------------------------------------------------- execution
def synth_weights(plan, consts, seed=1234):
    """one numpy array per weights.bin view: f32 ~ N(0, 1/sqrt(fan_in)), biases small; integer views from `consts`"""
    rng = np.random.default_rng(seed)
    out = {}
    for off, (kind, ln, shape) in plan["weights"].items():
        off = int(off)
        if kind == "weight_f32":
            n = int(np.prod(shape)) if shape else 1
            if len(shape) >= 2:
                a = rng.standard_normal(n) / np.sqrt(max(1, int(np.prod(shape[1:]))))
            else:
                a = rng.standard_normal(n) * 0.02
                if list(shape) == [4]:   # the bias of a box-distance head (left, top, right, bottom): positive, as trained distances are --
                    a = np.abs(a) + 0.5  # with zero-mean noise every synthetic box of a level is empty at once and nothing survives the filter
            if off in consts:
                a = np.asarray(consts[off], np.float64).reshape(-1)
            out[off] = a.astype(np.float32).reshape(shape)
        else:  # weight_i64 / weight_i64_f32 / weight_i32...: graph constants, must be supplied
            if off not in consts:
                raise SystemExit("integer constant at weights offset %d (shape %s) is needed: pass --const %d=..." % (off, shape, off))
            a = np.asarray(consts[off]).reshape(shape if shape else ())
            out[off] = a.astype(np.float32) if kind.endswith("_f32") else a.astype(np.int64)
    return out


from lele_amd.plan import Runner, fuse_sigmoid_mul, load_weights_bin, replan_lifted, weight_key  # noqa: E402  (the runner is shared with lele_amd.compiler plans)


def write_weights_bin(plan, weights, path):
    """the synthetic weights as a lele `<model>_weights.bin` (every view at its recorded byte offset, in its stored type), so
    that the native runner (lele_amd/lele_run) reads the very same values"""
    views = [(key, v if len(v) == 4 else [v[0], int(key), v[1], v[2]]) for key, v in plan["weights"].items()]   # format 2 / lifted
    blob = bytearray(max(off + ln for _key, (_kind, off, ln, _shape) in views))
    for key, (kind, off, ln, _shape) in views:
        a = np.asarray(weights[key if key in weights else int(key)])
        raw = (a.astype("<i8") if kind.startswith("weight_i64") else a.astype("<i4") if kind.startswith("weight_i32") else a.astype("<f4")).tobytes()
        assert len(raw) == ln, (off, kind, len(raw), ln)
        blob[off:off + ln] = raw
    open(path, "wb").write(bytes(blob))


def _yolo_anchor_grid(size=640, strides=(8, 16, 32)):
    """the detection tail's two geometry constants: anchor points [1, 2, A] = the cell centres (x + 0.5, y + 0.5) of the 80 x 80, 40 x 40 and
    20 x 20 maps in cell units, and the stride of each anchor [1, A] (A = 8400): boxes = (anchor -/+ distances) * stride"""
    pts, st = [], []
    for s_ in strides:
        n = size // s_
        ys, xs = np.meshgrid(np.arange(n, dtype=np.float32) + 0.5, np.arange(n, dtype=np.float32) + 0.5, indexing="ij")
        pts.append(np.stack([xs.reshape(-1), ys.reshape(-1)], 0))
        st.append(np.full(n * n, float(s_), np.float32))
    return np.concatenate(pts, 1)[None], np.concatenate(st)[None]


_ANCHORS, _STRIDES = _yolo_anchor_grid()

# YOLO26n-seg constants read from weights.bin (inferred from the graph: see --help); offset -> value
DEFAULT_CONSTS = {
    10644560: _ANCHORS,              # anchor points [1, 2, 8400]: `sub(anchors, lt)` / `add(anchors, rb)` (noise here would make every box empty)
    10951184: _STRIDES,              # stride per anchor [1, 8400]: `mul(xyxy, strides)`
    5445328: [1.0, 1.0, 2.0, 2.0],   # Resize scales (x2 nearest upsampling in the neck)
    7762768: [1, 64, 80, 80],        # proto Resize target size (features brought to the 80x80 level before fusion)
    10993152: [300],                 # TopK k (max detections)
    10993200: 80,                    # number of classes (flat index -> anchor = idx / 80, class = idx % 80)
}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    a = sub.add_parser("lift")
    a.add_argument("source")
    a.add_argument("-o", "--out", required=True)
    b = sub.add_parser("run")
    b.add_argument("plan")
    b.add_argument("--input-shape", default="1,3,640,640")
    b.add_argument("--runs", type=int, default=10)
    b.add_argument("--batch-runs", type=int, default=64, help="forwards per timed run (lele loops over images on the host)")
    b.add_argument("--const", default="")
    b.add_argument("--weights", default=None, help="a real <model>_weights.bin; default: synthetic weights")
    b.add_argument("--out", default=None)
    b.add_argument("--as-lifted", action="store_true", help="run the call sequence exactly as lifted (no sigmoid+mul -> silu peephole)")
    b.add_argument("--replan", action="store_true", help="re-assign buffers with this library's liveness allocator and fold conv2d+silu "
                   "(plan.replan_lifted); kept only if the outputs stay bit-identical")
    b.add_argument("--e2e", action="store_true", help="also time the whole application step on the device: u8 HWC image -> resize / normalise "
                   "(image.rs:62-111) -> the graph -> score filter, mask assembly and crop (image.rs:127-265); only the detections and the "
                   "u8 mask come back")
    b.add_argument("--streams", type=int, default=0, help="also replay the graph on N contexts (N HIP streams, one image each) at once")
    b.add_argument("--native", action="store_true", help="also run the plan with the native runner (lele_amd/lele_run) and compare")
    args = ap.parse_args()
    if args.cmd == "lift":
        plan = lift(args.source)
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(plan, open(args.out, "w"))
        ops = {}
        for s in plan["statements"]:
            if s["op"] == "call":
                ops[s["fn"]] = ops.get(s["fn"], 0) + 1
        print(json.dumps({"statements": len(plan["statements"]), "weights_views": len(plan["weights"]), "ops": ops}))
        return
    plan = json.load(open(args.plan))
    consts = dict(DEFAULT_CONSTS)
    for kv in filter(None, args.const.split(";")):
        k, v = kv.split("=")
        consts[int(k)] = json.loads(v)
    import lele_amd
    ctx = lele_amd._lib.Ctx(0)
    r = Runner(plan, load_weights_bin(plan, args.weights) if args.weights else synth_weights(plan, consts), ctx)
    shape = [int(v) for v in args.input_shape.split(",")]
    rng = np.random.default_rng(0)
    x = ctx.buf().upload(rng.uniform(0, 1, shape).astype(np.float32))  # section 8(d): uniform[0,1) images
    from lele_amd.tensor import TensorView
    inp = {plan["inputs"][-1]: TensorView(x)}
    r.shapes = {}
    outs = r.run(inp)
    shapes, r.shapes = r.shapes, None
    calls = r.calls
    reference_outputs = [o.numpy().copy() for o in outs]
    if not args.as_lifted:  # the bit-identical peephole lele's window matcher misses (plan.fuse_sigmoid_mul)
        fused = fuse_sigmoid_mul(plan, shapes)
        if len(fused["statements"]) != len(plan["statements"]):
            r2 = Runner(fused, r.raw, ctx)
            same = all(np.array_equal(a, b.numpy()) for a, b in zip(reference_outputs, r2.run(inp)))
            if not same:
                raise SystemExit("sigmoid+mul -> silu changed the outputs: refusing to use the fused plan")
            fused_away = len(plan["statements"]) - len(fused["statements"])
            plan, r = fused, r2
            calls_fused = calls - fused_away
        else:
            fused_away, calls_fused = 0, calls
    else:
        fused_away, calls_fused = 0, calls
    calls_replanned = None
    if args.replan and not args.as_lifted:
        re = replan_lifted(plan, shapes)
        raw2 = {weight_key([k, int(off), ln, shp]): r.raw[int(off)] for off, (k, ln, shp) in plan["weights"].items()}
        r3 = Runner(re, raw2, ctx)
        # conv2d + silu -> conv2d_silu is the same bits with the REPLICA of the reference's SiLU in the convolution's epilogue
        # (LELE_HIP_CONV_SILU_EXACT=1, read per call): both plans run under it for this check; the default epilogue (v_exp_f32 /
        # v_rcp_f32) is within 1e-5 of it, and what follows compares like with like (the re-planned graph against itself)
        had = os.environ.get("LELE_HIP_CONV_SILU_EXACT")
        os.environ["LELE_HIP_CONV_SILU_EXACT"] = "1"
        try:
            before = [o.numpy().copy() for o in r.run(inp)]
            same = all(np.array_equal(a, b.numpy()) for a, b in zip(before, r3.run(inp)))
        finally:
            if had is None:
                del os.environ["LELE_HIP_CONV_SILU_EXACT"]
            else:
                os.environ["LELE_HIP_CONV_SILU_EXACT"] = had
        if not same:
            raise SystemExit("replan_lifted changed the outputs: refusing to use the re-planned graph")
        r3.calls = 0
        reference_outputs = [o.numpy().copy() for o in r3.run(inp)]
        plan, r, calls_replanned = re, r3, r3.calls
    out_shapes = [list(o.shape) for o in outs]
    finite = all(bool(np.isfinite(o.numpy()).all()) for o in outs)
    for _ in range(2):
        r.run(inp)
    ctx.sync()
    r.profile = {}
    r.run(inp)
    prof = {k: round(1e3 * v, 3) for k, v in sorted(r.profile.items(), key=lambda kv: -kv[1])}
    r.profile = None
    eager = []
    for _ in range(args.runs):
        ctx.sync()
        t0 = time.perf_counter()
        r.run(inp)
        ctx.sync()
        eager.append(time.perf_counter() - t0)
    # the same sequence as one hipGraph, replayed batch_runs times per timed run (one image per forward, as lele does)
    graph_ms = None
    try:
        ctx.sync()
        ctx.graph_begin()
        r.run(inp)
        g = ctx.graph_end()
        g.launch()
        ctx.sync()
        ts = []
        for _ in range(args.runs):
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.batch_runs):
                g.launch()
            ctx.sync()
            ts.append(time.perf_counter() - t0)
        graph_ms = 1e3 * float(np.mean(ts)) / args.batch_runs
    except Exception as e:  # noqa: BLE001
        ctx.graph_abort()
        graph_ms = "capture failed: %s" % e
    rec = {"model": plan["source"], "input_shape": shape, "kernel_calls_per_forward": calls, "kernel_calls_after_silu_peephole": calls_fused,
           "kernel_calls_after_replan": calls_replanned, "buffers": len(plan["slots"]),
           "output_shapes": out_shapes,
           "finite": finite, "eager_ms_per_forward": round(1e3 * float(np.mean(eager)), 3), "graph_ms_per_forward": graph_ms,
           "images_per_s_graph": (round(1e3 / graph_ms, 1) if isinstance(graph_ms, float) else None),
           "per_op_ms_synced": prof,
           "note": "call sequence lifted from lele's generated source, synthetic weights, one image per forward"}
    if args.streams > 1 and isinstance(graph_ms, float):
        # N independent images in flight: one ctx (= one stream, one workspace, one recorded graph) each -- how a server would
        # keep the device busy with a graph compiled for N = 1 (lele's generated code loops over images on the host)
        ctxs = [ctx] + [lele_amd._lib.Ctx(0) for _ in range(args.streams - 1)]
        lanes = []
        xh = x.numpy() if hasattr(x, "numpy") else TensorView(x).numpy()
        for c in ctxs:
            rr = Runner(plan, r.raw, c)
            feed = {plan["inputs"][-1]: TensorView(c.buf().upload(xh))}
            first = rr.run(feed)
            if c is not ctx and not all(np.array_equal(a, b.numpy()) for a, b in zip(reference_outputs, first)):
                raise SystemExit("a second context computed different outputs")
            c.sync()
            c.graph_begin()
            rr.run(feed)
            lanes.append((c, c.graph_end(), rr, feed))
        for c, gph, _r, _f in lanes:
            gph.launch()
        for c, *_ in lanes:
            c.sync()
        ts = []
        for _ in range(args.runs):
            t0 = time.perf_counter()
            for _ in range(args.batch_runs // args.streams or 1):
                for c, gph, _r, _f in lanes:
                    gph.launch()
            for c, *_ in lanes:
                c.sync()
            ts.append(time.perf_counter() - t0)
        per_image = float(np.mean(ts)) / ((args.batch_runs // args.streams or 1) * args.streams)
        rec.update({"streams": args.streams, "streams_ms_per_image": round(1e3 * per_image, 4), "images_per_s_streams": round(1.0 / per_image, 1)})
    if args.e2e and isinstance(graph_ms, float) and shape == [1, 3, 640, 640]:
        from lele_amd import kernels as K
        ih, iw = 480, 640
        img = ctx.buf().upload(rng.integers(0, 256, (ih, iw, 3), dtype=np.uint8))
        K.image_preprocess(TensorView(img), 640, out=x.buf, ctx=ctx)   # writes the graph's input buffer
        ctx.sync()
        res = r.run(inp)
        db, cb_, mb = ctx.buf(), ctx.buf(), ctx.buf()
        e2e = {}
        for tag, thr in (("threshold_0.25", 0.25),):

            def app_step(thr=thr):
                K.image_preprocess(TensorView(img), 640, out=x.buf, ctx=ctx)
                g.launch()
                return K.yolo_seg_postprocess(res[0], res[1], iw, ih, thr, out_dets=db, out_count=cb_, out_mask=mb, ctx=ctx)
            dets, count, mask = app_step()
            n_det = int(count.raw().numpy()[0])
            ts = []
            for _ in range(args.runs):
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(args.batch_runs):
                    dets, count, mask = app_step()
                mask.raw().numpy()                               # the u8 mask and the detections are what leaves the device
                ctx.sync()
                ts.append(time.perf_counter() - t0)
            e2e[tag] = {"ms_per_image": round(1e3 * float(np.mean(ts)) / args.batch_runs, 4), "detections": n_det}
        rec.update({"e2e": e2e, "e2e_image": [ih, iw, 3],
                    "e2e_note": "u8 image -> preprocess -> graph -> postprocess, all on the device; with synthetic weights the boxes are "
                                "degenerate and none survives the filter (the mask pass still visits every pixel): the path is what is "
                                "timed, the post-processing's results are pinned by tests/test_app_steps.py"})
    if args.native:  # the same plan, the same weights, no Python: C++ runner over the C ABI
        import subprocess
        import tempfile
        exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lele_amd", "lele_run")
        with tempfile.TemporaryDirectory() as td:
            write_weights_bin(plan, r.raw, os.path.join(td, "w.bin"))
            json.dump(plan, open(os.path.join(td, "plan.json"), "w"))   # the plan as executed above (after the peephole)
            xin = x.numpy() if hasattr(x, "numpy") else TensorView(x).numpy()
            xin.tofile(os.path.join(td, "x.bin"))
            out = subprocess.run([exe, os.path.join(td, "plan.json"), os.path.join(td, "w.bin"), "--input",
                                  "%s=%s:f32:%s" % (plan["inputs"][-1], os.path.join(td, "x.bin"), ",".join(map(str, shape))), "--out",
                                  os.path.join(td, "o"), "--runs", str(args.batch_runs), "--graph"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if out.returncode == 0:
                nrec = json.loads(out.stdout.strip().splitlines()[-1])
                same = all(np.array_equal(np.fromfile(os.path.join(td, "o%d.bin" % k), np.float32).reshape(sh_), o.numpy())
                           for k, (sh_, o) in enumerate(zip(nrec["outputs"], r.run(inp))))
                rec.update({"native_eager_ms_per_forward": round(nrec["eager_ms"], 3), "native_graph_ms_per_forward": round(nrec["graph_ms"], 3),
                            "native_outputs_identical": bool(same), "native_kernel_calls": nrec["kernel_calls"]})
            else:
                rec["native_error"] = out.stderr.strip()[-400:]
    print(json.dumps(rec), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
