#!/usr/bin/env python3
"""SenseVoice-shaped encoder driven through the C ABI (BASELINE configs[2]/[3]).

The SenseVoiceSmall ONNX file is not in the reference tree, so this is the ASSUMED topology SURVEY.md documents from
lele's own sources (d=512, 4 heads x 128, FFN 2048, fused QKV 1536, FSMN depthwise conv k=11, input 560, vocab 25055,
70 SAN-M layers, 4 prompt tokens prepended) with synthetic weights drawn exactly as section 8(d) prescribes.  Every node
is one call of the operator library (lele_amd.kernels), i.e. the call sequence lele's generated Rust would make:

    front-end (PCM -> LFR) -> CMVN -> [LayerNorm -> fused_quantized_linear(QKV) -> split -> FSMN(conv1d) ->
    transpose -> matmul -> mul(scale) -> softmax -> matmul -> transpose -> fused_quantized_linear -> add -> add ->
    LayerNorm -> fused_quantized_linear(+ReLU) -> fused_quantized_linear -> add] x 70 -> LayerNorm -> CTC linear

Reported as lele's own harness does (examples/sensevoice/src/main.rs:198-237): mean wall time of >= 10 steady-state
runs after warm-up, divided by the audio duration = RTF; model-only and front-end + model.

    gpurun -- 'python tools/sensevoice_graph.py --out gpurun_out/sensevoice_r01.json'
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

D, HEADS, DH, FFN, VOCAB, FSMN_K = 512, 4, 128, 2048, 25055, 11


class QLinear:
    """weights of one fused_quantized_linear node (section 8(d) recipe)"""

    def __init__(self, rng, k, n, Weight):
        self.w = Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
        self.scale = Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32))
        self.zero = Weight(np.array([128.0], np.float32))
        self.bias = Weight((rng.standard_normal(n) * 0.02).astype(np.float32))

    def onnx(self, g, x, out, relu, tag):
        """the six-node form ONNX Runtime's dynamic quantisation emits, which lele's compiler fuses (patterns.rs:121-432)"""
        from lele_amd.compiler import onnx_pb as pb
        n = lambda s: "%s_%s" % (tag, s)  # noqa: E731
        g.initializer += [pb.Tensor(n("w"), self.w.arr.astype(np.uint8)), pb.Tensor(n("ws"), self.scale.arr),
                          pb.Tensor(n("wz"), np.array(int(self.zero.arr[0]), np.uint8)), pb.Tensor(n("b"), self.bias.arr)]
        lin = n("lin") if relu else out
        g.node += [pb.Node("DynamicQuantizeLinear", [x], [n("q"), n("s"), n("z")]), pb.Node("Mul", [n("s"), n("ws")], [n("cs")]),
                   pb.Node("MatMulInteger", [n("q"), n("w"), n("z"), n("wz")], [n("mm")]), pb.Node("Cast", [n("mm")], [n("mmf")], to=1),
                   pb.Node("Mul", [n("mmf"), n("cs")], [n("dq")]), pb.Node("Add", [n("dq"), n("b")], [lin])]
        if relu:
            g.node.append(pb.Node("Relu", [lin], [out]))


class Layer:
    def __init__(self, rng, d_in, Weight):
        ln = lambda n: (Weight((1 + 0.1 * rng.standard_normal(n)).astype(np.float32)),
                        Weight((0.1 * rng.standard_normal(n)).astype(np.float32)))
        self.d_in = d_in
        self.ln1, self.ln2 = ln(d_in), ln(D)
        self.qkv = QLinear(rng, d_in, 3 * D, Weight)
        self.out = QLinear(rng, D, D, Weight)
        self.ffn1 = QLinear(rng, D, FFN, Weight)
        self.ffn2 = QLinear(rng, FFN, D, Weight)
        self.fsmn = Weight((rng.standard_normal((D, 1, FSMN_K)) / np.sqrt(FSMN_K)).astype(np.float32))


# `damped=True`: the same section-8(d) draws with the projections INTO the residual stream scaled down, as a trained pre-LayerNorm
# transformer has them (GPT-2 initialises exactly those projections at 1 / sqrt(2 L)): the output projection and the second
# feed-forward linear of every layer x DAMP_BRANCH, the v third of the qkv projection (what the FSMN memory block adds to the stream)
# x DAMP_V, and layer 0's output projection -- which STARTS the stream, that layer has no residual input -- x DAMP_FIRST.  With the
# plain draws every branch is several times larger than the stream it is added to: each layer REPLACES its input, and a one-code
# flip of the dynamic u8 quantiser (what a last-bit difference between two correct implementations turns into) grows to the full
# quantisation noise of the branch within a few layers -- the logits of two bit-careful implementations then share 20 % of their
# arg-max ids and a graph-level comparison means nothing.  Damped, a branch is 0.05-0.25 of the stream (rms, measured on the
# oracle: stream ~40, attention 1, memory block 3, feed-forward 5) and such a flip decays instead: the whole 70-layer forward
# can be held against the oracle's (tests/test_graph_oracle.py; bench.py reports the agreement).  Kernel times do not depend on it.
DAMP_BRANCH, DAMP_V, DAMP_FIRST = (float(v) for v in os.environ.get("LELE_SV_DAMP", "0.05,0.25,2.0").split(","))


class Encoder:
    def __init__(self, ctx, layers=70, seed=1234, damped=False):
        import lele_amd
        from lele_amd import kernels as K
        from lele_amd._lib import Weight
        self.ctx, self.K = ctx, K
        rng = np.random.default_rng(seed)
        self.layers = [Layer(rng, 560 if i == 0 else D, Weight) for i in range(layers)]
        self.damped = bool(damped)
        if damped:   # in place, before any of the arrays is uploaded or packed
            for i, L in enumerate(self.layers):
                for lin, f in ((L.out, DAMP_FIRST if i == 0 else DAMP_BRANCH), (L.ffn2, DAMP_BRANCH)):
                    lin.scale.arr *= np.float32(f)
                    lin.bias.arr *= np.float32(f)
                L.qkv.scale.arr[2 * D:] *= np.float32(DAMP_V)
                L.qkv.bias.arr[2 * D:] *= np.float32(DAMP_V)
        self.ln_out = (Weight((1 + 0.1 * rng.standard_normal(D)).astype(np.float32)),
                       Weight((0.1 * rng.standard_normal(D)).astype(np.float32)))
        self.ctc = QLinear(rng, D, VOCAB, Weight)
        self.prompt = Weight((rng.standard_normal((1, 4, 560)) * 0.5).astype(np.float32))
        self.scale = Weight(np.array([DH ** -0.5], np.float32))
        # one workspace slot per distinct live value, as lele's compile-time buffer plan would assign
        self.ws = [ctx.buf() for _ in range(16)]

    def ql(self, x, p, relu, slot):
        return self.K.fused_quantized_linear(x, p.w, p.scale, p.zero, p.bias, relu, out=self.ws[slot], ctx=self.ctx)

    def embed(self, feats):
        """[B, T, 560] -> [B, T+4, 560]: the four prompt embeddings prepended"""
        K, ctx, ws = self.K, self.ctx, self.ws
        b = feats.shape[0]
        prompt = K.expand(self.prompt, [b, 4, 560], out=ws[15], ctx=ctx) if b > 1 else self.prompt
        return K.concat([prompt, feats], 1, out=ws[0], ctx=ctx)

    def layer(self, x, i, taps=None):
        """one SAN-M layer, one C-ABI call per node (the sequence lele's generated code would issue).  taps: optional dict that
        receives host copies of the intermediates (tests compare them with the oracle op by op)"""
        K, ctx, ws, L = self.K, self.ctx, self.ws, self.layers[i]
        b, t = x.shape[0], x.shape[1]
        xin = x
        xn = K.layer_norm(xin, L.ln1[0], L.ln1[1], -1, 1e-5, out=ws[1], ctx=ctx)
        qkv = self.ql(xn, L.qkv, False, 2)                                      # [B,T,1536]
        if taps is not None:
            taps.update(x=xin.numpy(), xn=xn.numpy(), qkv=qkv.numpy())
        q, k, v = K.split(qkv, 2, [D, D, D], outputs=[ws[3], ws[4], ws[5]], ctx=ctx)
        # FSMN memory: depthwise conv over time on v, plus v
        vt = K.transpose(v, [0, 2, 1], out=ws[6], ctx=ctx)                      # [B,512,T]
        mem = K.conv1d(vt, L.fsmn, None, [1], D, [FSMN_K // 2, FSMN_K // 2], [1], out=ws[7], ctx=ctx)
        mem = K.transpose(mem, [0, 2, 1], out=ws[6], ctx=ctx)                   # [B,T,512]
        mem = K.add(mem, v, out=ws[7], ctx=ctx)
        # attention
        qh = K.transpose(K.reshape(q, [b, t, HEADS, DH]), [0, 2, 1, 3], out=ws[8], ctx=ctx)   # [B,4,T,128]
        kh = K.transpose(K.reshape(k, [b, t, HEADS, DH]), [0, 2, 3, 1], out=ws[9], ctx=ctx)   # [B,4,128,T]
        vh = K.transpose(K.reshape(v, [b, t, HEADS, DH]), [0, 2, 1, 3], out=ws[10], ctx=ctx)
        sc = K.matmul(qh, kh, out=ws[3], ctx=ctx)                               # [B,4,T,T]
        if taps is not None:
            taps.update(mem=mem.numpy(), sc_raw=sc.numpy())
        sc = K.mul(sc, self.scale, out=ws[4], ctx=ctx)
        if taps is not None:
            taps.update(sc=sc.numpy())                                          # ws[4] is reused by the P.V product below
        pr = K.softmax(sc, -1, out=ws[3], ctx=ctx)
        av = K.matmul(pr, vh, out=ws[4], ctx=ctx)                               # [B,4,T,128]
        if taps is not None:
            taps.update(pr=pr.numpy(), av_heads=av.numpy())
        av = K.reshape(K.transpose(av, [0, 2, 1, 3], out=ws[5], ctx=ctx), [b, t, D])
        att = self.ql(av, L.out, False, 8)
        if taps is not None:
            taps.update(av=av.numpy(), att=att.numpy())
        slot_x = 11 if i % 2 == 0 else 12
        if L.d_in == D:
            att = K.add(att, mem, out=ws[9], ctx=ctx)
            x = K.add(att, xin, out=ws[slot_x], ctx=ctx)
        else:  # the first layer changes width (560 -> 512): no residual
            x = K.add(att, mem, out=ws[slot_x], ctx=ctx)
        xn = K.layer_norm(x, L.ln2[0], L.ln2[1], -1, 1e-5, out=ws[1], ctx=ctx)
        h = self.ql(xn, L.ffn1, True, 2)
        h2 = self.ql(h, L.ffn2, False, 3)
        y = K.add(x, h2, out=ws[13 if i % 2 == 0 else 14], ctx=ctx)
        if taps is not None:
            taps.update(x1=x.numpy(), xn2=xn.numpy(), h=h.numpy(), h2=h2.numpy(), y=y.numpy())
        return y

    def head(self, x):
        xn = self.K.layer_norm(x, self.ln_out[0], self.ln_out[1], -1, 1e-5, out=self.ws[1], ctx=self.ctx)
        return self.ql(xn, self.ctc, False, 2)

    def forward(self, feats, taps=None):
        """feats: [B, T, 560] device tensor (LFR + CMVN output) -> logits [B, T+4, VOCAB].  taps: {layer index: dict} to fill"""
        x = self.embed(feats)
        for i in range(len(self.layers)):
            x = self.layer(x, i, None if taps is None else taps.get(i))
        return self.head(x)


def encoder_arrays(enc):
    """the Encoder's weights as plain numpy arrays, in the layout oracle/sensevoice_ref.py consumes"""
    ql = lambda p: (p.w.arr, p.scale.arr, p.zero.arr, p.bias.arr)  # noqa: E731
    return {"prompt": enc.prompt.arr, "ln_out": (enc.ln_out[0].arr, enc.ln_out[1].arr), "ctc": ql(enc.ctc),
            "layers": [{"d_in": L.d_in, "ln1": (L.ln1[0].arr, L.ln1[1].arr), "ln2": (L.ln2[0].arr, L.ln2[1].arr), "qkv": ql(L.qkv),
                        "out": ql(L.out), "ffn1": ql(L.ffn1), "ffn2": ql(L.ffn2), "fsmn": L.fsmn.arr} for L in enc.layers]}


def encoder_onnx(enc, batch):
    """The same assumed topology as an ONNX model (bytes), built from the Encoder's weights: what a SenseVoice export
    with dynamically quantised linears looks like to lele's compiler.  `lele_amd.compiler` must turn it back into the
    call sequence of Encoder.forward."""
    from lele_amd.compiler import onnx_pb as pb
    g = pb.Graph([], [pb.ValueInfo("feats", pb.FLOAT, [batch, "t", 560])], [pb.ValueInfo("logits", pb.FLOAT, [batch, "t4", VOCAB])])
    I = g.initializer
    I += [pb.Tensor("prompt", enc.prompt.arr), pb.Tensor("att_scale", enc.scale.arr), pb.Tensor("shape_heads", np.array([0, 0, HEADS, DH], np.int64)),
          pb.Tensor("shape_merge", np.array([0, 0, D], np.int64)), pb.Tensor("split_qkv", np.array([D, D, D], np.int64))]
    if batch > 1:
        I.append(pb.Tensor("prompt_shape", np.array([batch, 4, 560], np.int64)))
        g.node.append(pb.Node("Expand", ["prompt", "prompt_shape"], ["prompt_b"]))
    g.node.append(pb.Node("Concat", ["prompt_b" if batch > 1 else "prompt", "feats"], ["x0"], axis=1))
    x = "x0"
    for i, L in enumerate(enc.layers):
        t = lambda s, i=i: "l%d_%s" % (i, s)  # noqa: E731
        I += [pb.Tensor(t("ln1_g"), L.ln1[0].arr), pb.Tensor(t("ln1_b"), L.ln1[1].arr), pb.Tensor(t("ln2_g"), L.ln2[0].arr),
              pb.Tensor(t("ln2_b"), L.ln2[1].arr), pb.Tensor(t("fsmn"), L.fsmn.arr)]
        g.node.append(pb.Node("LayerNormalization", [x, t("ln1_g"), t("ln1_b")], [t("xn")], axis=-1, epsilon=1e-5))
        L.qkv.onnx(g, t("xn"), t("qkv"), False, t("qkv"))
        g.node += [pb.Node("Split", [t("qkv"), "split_qkv"], [t("q"), t("k"), t("v")], axis=2),
                   pb.Node("Transpose", [t("v")], [t("vt")], perm=[0, 2, 1]),
                   pb.Node("Conv", [t("vt"), t("fsmn")], [t("memt")], group=D, pads=[FSMN_K // 2, FSMN_K // 2], kernel_shape=[FSMN_K]),
                   pb.Node("Transpose", [t("memt")], [t("mem0")], perm=[0, 2, 1]), pb.Node("Add", [t("mem0"), t("v")], [t("mem")])]
        for nm, perm in (("q", [0, 2, 1, 3]), ("k", [0, 2, 3, 1]), ("v", [0, 2, 1, 3])):
            g.node += [pb.Node("Reshape", [t(nm), "shape_heads"], [t(nm + "r")]), pb.Node("Transpose", [t(nm + "r")], [t(nm + "h")], perm=perm)]
        g.node += [pb.Node("MatMul", [t("qh"), t("kh")], [t("sc")]), pb.Node("Mul", [t("sc"), "att_scale"], [t("scs")]),
                   pb.Node("Softmax", [t("scs")], [t("pr")], axis=-1), pb.Node("MatMul", [t("pr"), t("vh")], [t("av")]),
                   pb.Node("Transpose", [t("av")], [t("avt")], perm=[0, 2, 1, 3]), pb.Node("Reshape", [t("avt"), "shape_merge"], [t("avm")])]
        L.out.onnx(g, t("avm"), t("att"), False, t("out"))
        if L.d_in == D:
            g.node += [pb.Node("Add", [t("att"), t("mem")], [t("am")]), pb.Node("Add", [t("am"), x], [t("x1")])]
        else:
            g.node.append(pb.Node("Add", [t("att"), t("mem")], [t("x1")]))
        g.node.append(pb.Node("LayerNormalization", [t("x1"), t("ln2_g"), t("ln2_b")], [t("x1n")], axis=-1, epsilon=1e-5))
        L.ffn1.onnx(g, t("x1n"), t("h1"), True, t("ffn1"))
        L.ffn2.onnx(g, t("h1"), t("h2"), False, t("ffn2"))
        g.node.append(pb.Node("Add", [t("x1"), t("h2")], [t("x2")]))
        x = t("x2")
    I += [pb.Tensor("lnf_g", enc.ln_out[0].arr), pb.Tensor("lnf_b", enc.ln_out[1].arr)]
    g.node.append(pb.Node("LayerNormalization", [x, "lnf_g", "lnf_b"], ["xf"], axis=-1, epsilon=1e-5))
    enc.ctc.onnx(g, "xf", "logits", False, "ctc")
    return pb.Model(g, opset=17).serialize()


def synth_pcm(batch, n, seed0=0):
    import bench
    return bench.synth_batch(batch, n, seed0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=70)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--configs", default="c3,c4")
    ap.add_argument("--no-graph", action="store_true", help="skip the hipGraph leg (rocprofv3 cannot trace captures)")
    ap.add_argument("--via-onnx", action="store_true", help="also build the model as ONNX, compile it with lele_amd.compiler and run the plan")
    ap.add_argument("--streams", type=int, default=0, help="with --via-onnx: also replay the compiled graph on N contexts (N HIP streams) at once")
    ap.add_argument("--compiled-only", action="store_true", help="profiling aid: only the compiled plan, eagerly, --runs times (use with rocprofv3)")
    args = ap.parse_args()
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd.features import Cmvn, SenseVoiceFrontend

    ctx = lele_amd._lib.Ctx(0)
    fe, cmvn = SenseVoiceFrontend(ctx=ctx), Cmvn(ctx=ctx)
    enc = Encoder(ctx, args.layers)
    results = []
    for name, batch, seconds in (("c3", 1, 30), ("c4", 32, 10)):
        if name not in args.configs.split(","):
            continue
        n = 16000 * seconds
        pcm = ctx.buf().upload(synth_pcm(batch, n))
        fbuf, cbufs = ctx.buf(), [ctx.buf() for _ in range(batch)]

        def frontend():
            f = fe.compute_batch(pcm, fbuf)                                    # [B, T, 560]
            return cmvn.compute(f, out=cbufs[0]) if batch > 1 else K.reshape(
                cmvn.compute(K.reshape(f, list(f.shape[1:])), out=cbufs[0]), [1] + list(f.shape[1:]))

        feats = frontend()
        if args.compiled_only:
            from lele_amd.compiler import compile_model
            from lele_amd.plan import Runner, load_weights_bin
            plan, blob = compile_model(encoder_onnx(enc, batch), "sensevoice_shaped")
            r = Runner(plan, load_weights_bin(plan, blob), ctx)
            for _ in range(2 + args.runs):
                r.run({"feats": feats})
            ctx.sync()
            print(json.dumps({"config": name, "compiled_only_forwards": 2 + args.runs, "kernel_calls_per_forward": r.calls // (2 + args.runs)}), flush=True)
            continue
        for _ in range(2):  # warm-up: uploads and pre-packs every weight once
            logits = enc.forward(feats)
        ctx.sync()
        t_model, t_all = [], []
        for _ in range(args.runs):
            ctx.sync()
            t0 = time.perf_counter()
            logits = enc.forward(feats)
            ctx.sync()
            t_model.append(time.perf_counter() - t0)
        for _ in range(args.runs):
            ctx.sync()
            t0 = time.perf_counter()
            logits = enc.forward(frontend())
            ctx.sync()
            t_all.append(time.perf_counter() - t0)
        # the same call sequence recorded once and replayed as one hipGraph launch (lele_hip_graph_*)
        t_graph = [float("nan")] if args.no_graph else []
        ctx.sync()
        if not args.no_graph:
            ctx.graph_begin()
            logits = enc.forward(feats)
            graph = ctx.graph_end()
            graph.launch()
            ctx.sync()
        for _ in range(0 if args.no_graph else args.runs):
            ctx.sync()
            t0 = time.perf_counter()
            graph.launch()
            ctx.sync()
            t_graph.append(time.perf_counter() - t0)
        onnx_rec = {}
        if args.via_onnx:  # ONNX bytes -> compiled plan -> the same call sequence; logits must be identical
            from lele_amd.compiler import compile_model
            from lele_amd.plan import Runner, load_weights_bin
            from lele_amd.tensor import TensorView as TensorViewOf
            t0 = time.perf_counter()
            data = encoder_onnx(enc, batch)
            plan, blob = compile_model(data, "sensevoice_shaped")
            t_compile = time.perf_counter() - t0
            r = Runner(plan, load_weights_bin(plan, blob), ctx)
            want = enc.forward(feats).numpy()
            got = r.run({"feats": feats})[0]
            same = bool(np.array_equal(got.numpy(), want))
            ctx.sync()
            ctx.graph_begin()
            r.run({"feats": feats})
            g2 = ctx.graph_end()
            g2.launch()
            ctx.sync()
            tg = []
            for _ in range(args.runs):
                ctx.sync()
                t0 = time.perf_counter()
                g2.launch()
                ctx.sync()
                tg.append(time.perf_counter() - t0)
            fn_count = {}
            for st in plan["statements"]:
                if st["op"] == "call":
                    fn_count[st["fn"]] = fn_count.get(st["fn"], 0) + 1
            onnx_rec = {"onnx_bytes": len(data), "onnx_nodes": sum(1 for _ in __import__("lele_amd.compiler.onnx_pb", fromlist=["x"]).load(data).graph.node),
                        "plan_statements": len(plan["statements"]), "plan_slots": len(plan["slots"]), "weights_bin_bytes": len(blob),
                        "compile_s": round(t_compile, 2), "plan_calls": fn_count, "compiled_logits_identical": same,
                        "compiled_graph_ms": round(1e3 * float(np.mean(tg)), 3)}
            if args.streams > 1:  # N independent requests in flight: one ctx (= one stream, one workspace, one graph) each
                import lele_amd as _la
                weights = load_weights_bin(plan, blob)
                ctxs = [ctx] + [_la._lib.Ctx(0) for _ in range(args.streams - 1)]
                graphs = []
                fh = feats.numpy()
                for c in ctxs:
                    rr = Runner(plan, weights, c)
                    fx = TensorViewOf(c.buf().upload(fh))
                    rr.run({"feats": fx})
                    c.sync()
                    c.graph_begin()
                    rr.run({"feats": fx})
                    graphs.append((c, c.graph_end(), rr, fx))
                for c, gph, _r, _f in graphs:
                    gph.launch()
                for c, *_ in graphs:
                    c.sync()
                ts = []
                for _ in range(args.runs):
                    t0 = time.perf_counter()
                    for c, gph, _r, _f in graphs:
                        gph.launch()
                    for c, *_ in graphs:
                        c.sync()
                    ts.append(time.perf_counter() - t0)
                onnx_rec.update({"streams": args.streams, "streams_ms_per_round": round(1e3 * float(np.mean(ts)), 3),
                                 "streams_ms_per_forward": round(1e3 * float(np.mean(ts)) / args.streams, 3)})
            # the same plan through the native runner (lele_amd/lele_run: C++ over the C ABI, no Python on the serving path)
            exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lele_amd", "lele_run")
            if os.path.exists(exe):
                import subprocess
                import tempfile
                with tempfile.TemporaryDirectory() as td:
                    json.dump(plan, open(os.path.join(td, "m_plan.json"), "w"))
                    open(os.path.join(td, "m_weights.bin"), "wb").write(blob)
                    fh = feats.numpy()
                    fh.tofile(os.path.join(td, "feats.bin"))
                    out = subprocess.run([exe, os.path.join(td, "m_plan.json"), os.path.join(td, "m_weights.bin"), "--input",
                                          "feats=%s:f32:%s" % (os.path.join(td, "feats.bin"), ",".join(map(str, fh.shape))), "--out",
                                          os.path.join(td, "o"), "--runs", str(args.runs), "--graph"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
                    if out.returncode == 0:
                        nrec = json.loads(out.stdout.strip().splitlines()[-1])
                        native_logits = np.fromfile(os.path.join(td, "o0.bin"), np.float32).reshape(nrec["outputs"][0])
                        onnx_rec.update({"native_eager_ms": round(nrec["eager_ms"], 3), "native_graph_ms": round(nrec["graph_ms"], 3),
                                         "native_logits_identical": bool(np.array_equal(native_logits, want))})
                    else:
                        onnx_rec["native_error"] = out.stderr.strip()[-300:]
            logits = enc.forward(feats)
        audio = batch * seconds
        lg = logits.numpy()
        rec = {"config": name, "batch": batch, "seconds_per_utterance": seconds, "layers": args.layers,
               "tokens": int(lg.shape[1]), "logits_shape": list(lg.shape), "finite": bool(np.isfinite(lg).all()),
               "model_ms": round(1e3 * float(np.mean(t_model)), 3), "frontend_plus_model_ms": round(1e3 * float(np.mean(t_all)), 3),
               "rtf_model": round(float(np.mean(t_model)) / audio, 6), "rtf_total": round(float(np.mean(t_all)) / audio, 6),
               "model_graph_ms": round(1e3 * float(np.mean(t_graph)), 3),
               "rtf_model_graph": round(float(np.mean(t_graph)) / audio, 6),
               "note": "assumed topology, synthetic weights, every node a separate C-ABI call issued from Python"}
        rec.update(onnx_rec)
        print(json.dumps(rec), flush=True)
        results.append(rec)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
