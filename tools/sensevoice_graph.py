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


class Encoder:
    def __init__(self, ctx, layers=70, seed=1234):
        import lele_amd
        from lele_amd import kernels as K
        from lele_amd._lib import Weight
        self.ctx, self.K = ctx, K
        rng = np.random.default_rng(seed)
        self.layers = [Layer(rng, 560 if i == 0 else D, Weight) for i in range(layers)]
        self.ln_out = (Weight((1 + 0.1 * rng.standard_normal(D)).astype(np.float32)),
                       Weight((0.1 * rng.standard_normal(D)).astype(np.float32)))
        self.ctc = QLinear(rng, D, VOCAB, Weight)
        self.prompt = Weight((rng.standard_normal((1, 4, 560)) * 0.5).astype(np.float32))
        self.scale = Weight(np.array([DH ** -0.5], np.float32))
        # one workspace slot per distinct live value, as lele's compile-time buffer plan would assign
        self.ws = [ctx.buf() for _ in range(16)]

    def ql(self, x, p, relu, slot):
        return self.K.fused_quantized_linear(x, p.w, p.scale, p.zero, p.bias, relu, out=self.ws[slot], ctx=self.ctx)

    def forward(self, feats):
        """feats: [B, T, 560] device tensor (LFR + CMVN output) -> logits [B, T+4, VOCAB]"""
        K, ctx, ws = self.K, self.ctx, self.ws
        b = feats.shape[0]
        prompt = K.expand(self.prompt, [b, 4, 560], out=ws[15], ctx=ctx) if b > 1 else self.prompt
        x = K.concat([prompt, feats], 1, out=ws[0], ctx=ctx)
        t = x.shape[1]
        for i, L in enumerate(self.layers):
            xin = x
            xn = K.layer_norm(xin, L.ln1[0], L.ln1[1], -1, 1e-5, out=ws[1], ctx=ctx)
            qkv = self.ql(xn, L.qkv, False, 2)                                      # [B,T,1536]
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
            sc = K.mul(sc, self.scale, out=ws[4], ctx=ctx)
            pr = K.softmax(sc, -1, out=ws[3], ctx=ctx)
            av = K.matmul(pr, vh, out=ws[4], ctx=ctx)                               # [B,4,T,128]
            av = K.reshape(K.transpose(av, [0, 2, 1, 3], out=ws[5], ctx=ctx), [b, t, D])
            att = self.ql(av, L.out, False, 8)
            slot_x = 11 if i % 2 == 0 else 12
            if L.d_in == D:
                att = K.add(att, mem, out=ws[9], ctx=ctx)
                x = K.add(att, xin, out=ws[slot_x], ctx=ctx)
            else:  # the first layer changes width (560 -> 512): no residual
                x = K.add(att, mem, out=ws[slot_x], ctx=ctx)
            xn = K.layer_norm(x, L.ln2[0], L.ln2[1], -1, 1e-5, out=ws[1], ctx=ctx)
            h = self.ql(xn, L.ffn1, True, 2)
            h = self.ql(h, L.ffn2, False, 3)
            x = K.add(x, h, out=ws[13 if i % 2 == 0 else 14], ctx=ctx)
        xn = K.layer_norm(x, self.ln_out[0], self.ln_out[1], -1, 1e-5, out=ws[1], ctx=ctx)
        return self.ql(xn, self.ctc, False, 2)


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
        audio = batch * seconds
        lg = logits.numpy()
        rec = {"config": name, "batch": batch, "seconds_per_utterance": seconds, "layers": args.layers,
               "tokens": int(lg.shape[1]), "logits_shape": list(lg.shape), "finite": bool(np.isfinite(lg).all()),
               "model_ms": round(1e3 * float(np.mean(t_model)), 3), "frontend_plus_model_ms": round(1e3 * float(np.mean(t_all)), 3),
               "rtf_model": round(float(np.mean(t_model)) / audio, 6), "rtf_total": round(float(np.mean(t_all)) / audio, 6),
               "model_graph_ms": round(1e3 * float(np.mean(t_graph)), 3),
               "rtf_model_graph": round(float(np.mean(t_graph)) / audio, 6),
               "note": "assumed topology, synthetic weights, every node a separate C-ABI call issued from Python"}
        print(json.dumps(rec), flush=True)
        results.append(rec)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(results, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
