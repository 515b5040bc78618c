#!/usr/bin/env python3
"""Per-operator roofline table: every hot operator at the shapes the BASELINE configs use,
timed on the device through the C ABI with inputs resident in HBM, reported against its governing roofline.

    gpurun -- 'python tools/microbench.py --out gpurun_out/microbench_r01.json'
    cp gpurun_out/microbench_r01.json profiles/r01_microbench.json

Timing: HIP events on the ctx stream (lele_hip_timer_start/stop) around `iters` back-to-back calls after 3 warm-up
calls.  flops / bytes are ALGORITHMIC (2*M*N*K; each operand and the result counted once)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PEAK = {"mfma_f32": 157.3e12, "mfma_i8": 3.944e15, "hbm": 8.0e12}  # MI355X_MICROARCH.md (dense peaks; HBM spec)
PEAK_BF16 = 2.5e15  # dense bf16 MFMA; a six-term split-bf16 product of f32 operands runs at a sixth of it (conv_window_p_kernel)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--only", default="", help="comma-separated section names: matmul,quant,conv,conv1d,eltwise,c4,rnn,misc,frontend")
    args = ap.parse_args()
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import Weight

    ctx = lele_amd._lib.Ctx(0)
    rng = np.random.default_rng(1234)
    rows = []

    def dev(a):
        return ctx.buf().upload(np.ascontiguousarray(a))

    def f32(*shape, scale=1.0):
        return (rng.standard_normal(shape) * scale).astype(np.float32)

    def timeit(fn, iters):
        out = ctx.buf()
        for _ in range(3):
            fn(out)
        ctx.sync()
        ctx.timer_start()
        for _ in range(iters):
            fn(out)
        return ctx.timer_stop() / iters

    def record(op, shape, ms, flops, nbytes, bound, note=""):
        r = {"op": op, "shape": shape, "ms": round(ms, 5), "bound": bound}
        if flops:
            r["TFLOP/s" if bound != "mfma_i8" else "TOP/s"] = round(flops / ms / 1e9, 2)
        if nbytes:
            r["GB/s"] = round(nbytes / ms / 1e6, 1)
        if bound in ("mfma_f32", "mfma_i8"):
            r["frac"] = round(flops / (ms * 1e-3) / PEAK[bound], 4)
            if op.startswith("conv2d"):  # the batched 3 x 3 / 1 x 1 routes multiply in six-term split-bf16: the other peak, too
                r["frac_of_split_bf16_peak"] = round(flops / (ms * 1e-3) / (PEAK_BF16 / 6.0), 4)
        elif bound == "hbm":
            r["frac"] = round(nbytes / (ms * 1e-3) / PEAK["hbm"], 4)
        if note:
            r["note"] = note
        rows.append(r)
        print(json.dumps(r), flush=True)

    it = 5 if args.quick else 20
    nb = 8 if args.quick else 64
    only = [t for t in args.only.split(",") if t]

    def want(name):
        return not only or name in only

    # ---- f32 MatMul (a6/a7): SenseVoice attention shapes, bench shapes of benches/kernels.rs:310-315, one large square
    if want('matmul'):
        for b, m, k, n in [(4, 504, 128, 504), (4, 504, 504, 128), (1, 504, 512, 2048), (1, 512, 512, 512),
                           (2, 400, 32, 400), (1, 4096, 4096, 4096)]:
            a, bb = dev(f32(b, m, k)), dev(f32(b, k, n))
            ms = timeit(lambda o: K.matmul(a, bb, out=o, ctx=ctx), it)
            record("matmul", [b, m, k, n], ms, 2.0 * b * m * k * n, 4.0 * b * (m * k + k * n + m * n), "mfma_f32")
        a, bb, bias = dev(f32(504, 512)), dev(f32(512, 2048)), dev(f32(2048))
        ms = timeit(lambda o: K.matmul_fused_add(a, bb, bias, out=o, ctx=ctx), it)
        record("matmul_fused_add", [504, 512, 2048], ms, 2.0 * 504 * 512 * 2048, 4.0 * (504 * 512 + 512 * 2048 + 504 * 2048),
               "mfma_f32")

    # ---- quantized linear (a17): SenseVoice encoder shapes at M = 504 tokens
    if want('quant'):
        for m, k, n in [(504, 560, 1536), (504, 512, 1536), (504, 512, 512), (504, 512, 2048), (504, 2048, 512),
                        (504, 512, 25055), (8064, 512, 2048)]:
            x = dev(f32(1, m, k))
            w = Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
            ws = Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32))
            wz = Weight(np.array([128.0], np.float32))
            bs = Weight(f32(n, scale=0.02))
            ms = timeit(lambda o: K.fused_quantized_linear(x, w, ws, wz, bs, False, out=o, ctx=ctx), it)
            record("fused_quantized_linear", [m, k, n], ms, 2.0 * m * k * n, 4.0 * m * k + k * n + 8.0 * n + 4.0 * m * n,
                   "mfma_i8", "dynamic quantisation of the activations included")

    # ---- Conv2d (a9): Yolo26n-seg shapes at batch 64 (C5), conv_transpose (a10)
    if want('conv'):
        for c, h, oc, k, s, g, act in [(64, 160, 64, 3, 1, 1, "silu"), (3, 640, 16, 3, 2, 1, "silu"), (128, 80, 128, 1, 1, 1, "silu"),
                                       (256, 20, 256, 3, 1, 1, "silu"), (128, 40, 128, 3, 1, 128, None)]:
            x = dev(f32(nb, c, h, h))
            w = Weight(f32(oc, c // g, k, k, scale=0.1))
            bs = Weight(f32(oc))
            p = k // 2
            fn = K.conv2d_silu if act == "silu" else K.conv2d
            ms = timeit(lambda o: fn(x, w, bs, [1, 1], g, [p, p, p, p], [s, s], out=o, ctx=ctx), max(3, it // 4))
            oh = (h + 2 * p - k) // s + 1
            fl = 2.0 * nb * oc * (c // g) * k * k * oh * oh
            by = 4.0 * nb * (c * h * h + oc * oh * oh)
            record("conv2d" + ("_silu" if act else ""), [nb, c, h, h, oc, k, s, g], ms, fl, by, "hbm" if g > 1 else "mfma_f32")
        x = dev(f32(nb, 64, 80, 80))
        w = Weight(f32(64, 64, 2, 2, scale=0.1))
        ms = timeit(lambda o: K.conv_transpose(x, w, None, [1, 1], 1, [0, 0, 0, 0], [2, 2], out=o, ctx=ctx), 3)
        record("conv_transpose", [nb, 64, 80, 80, 64, 2, 2], ms, 2.0 * nb * 64 * 64 * 4 * 80 * 80,
               4.0 * nb * 64 * (80 * 80 + 160 * 160), "mfma_f32", "one implicit GEMM per output phase")

    # ---- Conv1d (a8): FSMN depthwise k=11 and the Silero STFT-as-conv
    if want('conv1d'):
        x, w = dev(f32(1, 512, 514)), Weight(f32(512, 1, 11))
        ms = timeit(lambda o: K.conv1d(x, w, None, [1], 512, [0, 0], [1], out=o, ctx=ctx), it)
        record("conv1d_depthwise", [1, 512, 514, 11], ms, 2.0 * 512 * 504 * 11, 4.0 * 512 * (514 + 504), "hbm")
        x, w = dev(f32(1, 1, 640)), Weight(f32(258, 1, 256))
        ms = timeit(lambda o: K.conv1d(x, w, None, [1], 1, [0, 0], [128], out=o, ctx=ctx), it)
        record("conv1d_stft", [1, 1, 640, 258, 256], ms, 2.0 * 258 * 256 * 4, 0, "latency", "Silero chunk; launch-latency bound")

    # ---- normalisation / activations / data movement (a13-a16, a19): HBM-bound
    if want('eltwise'):
        x, gmm, bta = dev(f32(1, 504, 512)), dev(f32(512)), dev(f32(512))
        ms = timeit(lambda o: K.layer_norm(x, gmm, bta, -1, 1e-5, out=o, ctx=ctx), it)
        record("layer_norm", [1, 504, 512], ms, 0, 8.0 * 504 * 512, "hbm", "1 MB problem: launch-latency dominated")
        x, gmm, bta = dev(f32(64, 504, 512)), dev(f32(512)), dev(f32(512))
        ms = timeit(lambda o: K.layer_norm(x, gmm, bta, -1, 1e-5, out=o, ctx=ctx), it)
        record("layer_norm", [64, 504, 512], ms, 0, 8.0 * 64 * 504 * 512, "hbm")
        x = dev(f32(1, 4, 504, 504))
        ms = timeit(lambda o: K.softmax(x, -1, out=o, ctx=ctx), it)
        record("softmax", [1, 4, 504, 504], ms, 0, 8.0 * 4 * 504 * 504, "hbm")
        x = dev(f32(64, 4, 504, 504))
        ms = timeit(lambda o: K.softmax(x, -1, out=o, ctx=ctx), it)
        record("softmax", [64, 4, 504, 504], ms, 0, 8.0 * 64 * 4 * 504 * 504, "hbm")
        x = dev(f32(nb, 16, 320, 320))
        ms = timeit(lambda o: K.silu(x, out=o, ctx=ctx), it)
        record("silu", [nb, 16, 320, 320], ms, 0, 8.0 * nb * 16 * 320 * 320, "hbm")
        y = dev(f32(nb, 16, 320, 320))
        ms = timeit(lambda o: K.add(x, y, out=o, ctx=ctx), it)
        record("add", [nb, 16, 320, 320], ms, 0, 12.0 * nb * 16 * 320 * 320, "hbm")
        x = dev(f32(64, 504, 4, 128))
        ms = timeit(lambda o: K.transpose(x, [0, 2, 1, 3], out=o, ctx=ctx), it)
        record("transpose_0213", [64, 504, 4, 128], ms, 0, 8.0 * 64 * 504 * 512, "hbm")
        x = dev(f32(nb, 128, 20, 20))
        ms = timeit(lambda o: K.max_pool2d(x, [5, 5], [1, 1], [2, 2, 2, 2], out=o, ctx=ctx), it)
        record("max_pool2d_5x5", [nb, 128, 20, 20], ms, 0, 8.0 * nb * 128 * 400, "hbm")

    # ---- one SenseVoice encoder layer at the batched C4 shapes (32 utterances x 171 tokens = 5472 rows)
    if want('c4'):
        B, T, H, DH, D = 32, 171, 4, 128, 512
        a, bb = dev(f32(B * H, T, DH)), dev(f32(B * H, DH, T))
        ms = timeit(lambda o: K.matmul(a, bb, out=o, ctx=ctx), it)
        record("matmul_qk", [B * H, T, DH, T], ms, 2.0 * B * H * T * DH * T, 4.0 * B * H * (2 * T * DH + T * T), "mfma_f32")
        a, bb = dev(f32(B * H, T, T)), dev(f32(B * H, T, DH))
        ms = timeit(lambda o: K.matmul(a, bb, out=o, ctx=ctx), it)
        record("matmul_pv", [B * H, T, T, DH], ms, 2.0 * B * H * T * DH * T, 4.0 * B * H * (2 * T * DH + T * T), "mfma_f32")
        for m, k, n in [(B * T, 512, 1536), (B * T, 512, 512), (B * T, 512, 2048), (B * T, 2048, 512)]:
            x = dev(f32(1, m, k))
            w = Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
            ws = Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32))
            wz = Weight(np.array([128.0], np.float32))
            bs = Weight(f32(n, scale=0.02))
            ms = timeit(lambda o: K.fused_quantized_linear(x, w, ws, wz, bs, False, out=o, ctx=ctx), it)
            record("fused_quantized_linear", [m, k, n], ms, 2.0 * m * k * n, 4.0 * m * k + k * n + 8.0 * n + 4.0 * m * n, "mfma_i8")
        x, w = dev(f32(B, D, T)), Weight(f32(D, 1, 11))
        ms = timeit(lambda o: K.conv1d(x, w, None, [1], D, [5, 5], [1], out=o, ctx=ctx), it)
        record("conv1d_depthwise", [B, D, T, 11], ms, 2.0 * B * D * T * 11, 8.0 * B * D * T, "hbm")
        x = dev(f32(B, T, D))
        ms = timeit(lambda o: K.transpose(x, [0, 2, 1], out=o, ctx=ctx), it)
        record("transpose_021", [B, T, D], ms, 0, 8.0 * B * T * D, "hbm")
        x4 = dev(f32(B, T, H, DH))
        ms = timeit(lambda o: K.transpose(x4, [0, 2, 1, 3], out=o, ctx=ctx), it)
        record("transpose_0213", [B, T, H, DH], ms, 0, 8.0 * B * T * D, "hbm")
        ms = timeit(lambda o: K.transpose(x4, [0, 2, 3, 1], out=o, ctx=ctx), it)
        record("transpose_0231", [B, T, H, DH], ms, 0, 8.0 * B * T * D, "hbm")
        qkv = dev(f32(B, T, 3 * D))
        outs = [ctx.buf() for _ in range(3)]
        ms = timeit(lambda o: K.split(qkv, 2, [D, D, D], outputs=outs, ctx=ctx), it)
        record("split3", [B, T, 3 * D], ms, 0, 8.0 * B * T * 3 * D, "hbm")
        y = dev(f32(B, T, D))
        ms = timeit(lambda o: K.add(x, y, out=o, ctx=ctx), it)
        record("add", [B, T, D], ms, 0, 12.0 * B * T * D, "hbm")
        sc, one = dev(f32(B, H, T, T)), dev(np.array([0.088], np.float32))
        ms = timeit(lambda o: K.mul(sc, one, out=o, ctx=ctx), it)
        record("mul_scalar", [B, H, T, T], ms, 0, 8.0 * B * H * T * T, "hbm")
        ms = timeit(lambda o: K.softmax(sc, -1, out=o, ctx=ctx), it)
        record("softmax", [B, H, T, T], ms, 0, 8.0 * B * H * T * T, "hbm")
        gmm, bta = dev(f32(D)), dev(f32(D))
        ms = timeit(lambda o: K.layer_norm(x, gmm, bta, -1, 1e-5, out=o, ctx=ctx), it)
        record("layer_norm", [B, T, D], ms, 0, 8.0 * B * T * D, "hbm")

    # ---- LSTM / GRU (a11/a12): latency-bound, microseconds per step
    if want('rnn'):
        for T in (1, 175):
            x, w, r, b = dev(f32(T, 1, 128)), Weight(f32(1, 512, 128, scale=0.1)), Weight(f32(1, 512, 128, scale=0.1)), Weight(f32(1, 1024))
            out3 = [None]

            def run(_o):
                out3[0] = K.lstm(x, w, r, b, None, None, None, ctx=ctx)
            ms = timeit(run, it)
            record("lstm_H128", [T, 1, 128], ms, 0, 0, "latency", "%.2f us per step" % (ms * 1e3 / T))

    # ---- front-end pieces outside the fused kernel: cmvn, stft
    if want('misc'):
        x = dev(f32(500, 560))
        ms = timeit(lambda o: lele_amd.features.Cmvn(ctx=ctx).compute(x, out=o), it)
        record("cmvn", [500, 560], ms, 0, 8.0 * 500 * 560, "hbm", "single utterance: launch-latency dominated")

    # ---- the front-end outside its fused default (pipeline.rs:38-42: n_fft = 1024 when the frame is longer than 512 samples): the
    # composed path -- frame means, pre-emphasis / window, the generic radix-2 FFT, sparse mel / log, LFR gather -- against the fused
    # kernel on the same audio length ("bit-exact but unprofiled")
    if want('misc') or want('frontend'):
        from lele_amd.features import FeatureConfig, SenseVoiceFrontend
        for label, cfg, sr in (("fused default 16 kHz / 25 ms / n_fft 512", FeatureConfig(), 16000),
                               ("generic 32 kHz / 25 ms / n_fft 1024", FeatureConfig(sample_rate=32000), 32000),
                               ("generic 16 kHz / 20 ms / n_fft 512", FeatureConfig(frame_length_ms=20.0), 16000)):
            nb_, secs = (64 if args.quick else 256), 30
            pcm = dev((rng.standard_normal((nb_, sr * secs)) * 0.1).astype(np.float32))
            fe = SenseVoiceFrontend(cfg, ctx=ctx)
            ms = timeit(lambda o: fe.compute_batch(pcm, out=o), max(3, it // 4))
            t_rows = fe.compute_batch(pcm).shape[1]
            record("frontend", [nb_, sr * secs, label], ms, 0, 4.0 * nb_ * (sr * secs + t_rows * cfg.n_mels * cfg.lfr_m), "hbm",
                   "algorithmic bytes: PCM read once, LFR written once")

    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump({"peaks": PEAK, "rows": rows}, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
