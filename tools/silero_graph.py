#!/usr/bin/env python3
"""Config C1: a Silero-VAD-shaped network streamed chunk by chunk -- 175 chunks of 512 samples (5.6 s
at 16 kHz), LSTM state carried on the device from one chunk to the next.

The ONNX file is built here the way Silero's is laid out (one network per sample rate under an `If` on `sr`; STFT as a
strided convolution with a 258 x 256 basis, magnitude, four k=3 convolution blocks 129 -> 128 -> 64 -> 64 -> 128, one LSTM
step with H = 128, a 1x1 convolution + sigmoid head), exported from torch, compiled by lele_amd.compiler and replayed as a
hipGraph.  The topology is assumed (the model file is not in the reference tree) and the weights are random: what is measured
is the per-chunk latency of the path -- the LSTM/GRU rows are latency-bound and carry no roofline claim.

Two loops are timed: `streaming` reads the speech probability on the host after every chunk (as examples/silero/src/main.rs
does for its segment state machine: one stream synchronisation per chunk) and `batched` only synchronises at the end."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHUNK, CONTEXT, HID = 512, 64, 128


def build_onnx():
    import torch
    from lele_amd.compiler import onnx_pb as pb
    from tests.onnx_util import export, splice_if

    class Net(torch.nn.Module):
        def __init__(self, win, seed):
            super().__init__()
            torch.manual_seed(seed)
            self.win = win
            self.stft = torch.nn.Conv1d(1, 2 * (win // 2 + 1), win, stride=win // 2, bias=False)
            ch = [win // 2 + 1, 128, 64, 64, 128]
            self.enc = torch.nn.ModuleList(torch.nn.Conv1d(ch[i], ch[i + 1], 3, stride=(1, 2, 2, 1)[i], padding=1) for i in range(4))
            self.lstm = torch.nn.LSTM(128, HID)
            self.head = torch.nn.Conv1d(HID, 1, 1)

        def forward(self, x, h0, c0):                       # x [1, CONTEXT + CHUNK]; state [1, 1, HID]
            s = self.stft(x.unsqueeze(1))
            half = self.win // 2 + 1
            y = torch.sqrt(s[:, :half] ** 2 + s[:, half:] ** 2)
            for conv in self.enc:
                y = torch.relu(conv(y))
            y, (hn, cn) = self.lstm(y.permute(2, 0, 1), (h0, c0))
            p = torch.sigmoid(self.head(torch.relu(y).permute(1, 2, 0)))
            return p.mean(dim=2), hn, cn

    ex = (torch.zeros(1, CONTEXT + CHUNK), torch.zeros(1, 1, HID), torch.zeros(1, 1, HID))
    kw = dict(opset=17, input_names=("x", "h0", "c0"), output_names=("prob", "hn", "cn"))
    parts = [export(Net(256, 1).eval(), ex, **kw), export(Net(128, 2).eval(), ex, **kw)]
    return splice_if(parts[0], parts[1], {"x", "h0", "c0"}, [pb.Node("Equal", ["sr", "sr16k"], ["is16k"])], "is16k",
                     [pb.ValueInfo("x", pb.FLOAT, [1, CONTEXT + CHUNK]), pb.ValueInfo("sr", pb.INT64, [1]),
                      pb.ValueInfo("h0", pb.FLOAT, [1, 1, HID]), pb.ValueInfo("c0", pb.FLOAT, [1, 1, HID])],
                     [pb.Tensor("sr16k", np.array([16000], np.int64))])


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--chunks", type=int, default=175)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--out", default=None)
    ap.add_argument("--streams", type=int, default=0, help="also run N independent audio streams side by side (one context each)")
    ap.add_argument("--bind-sr", action="store_true", help="fix sr = 16000 at compile time: the `If` is inlined, the 8 kHz network dropped")
    args = ap.parse_args()
    import lele_amd
    from lele_amd import apps, kernels as K
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.tensor import TensorView

    data = build_onnx()
    t0 = time.perf_counter()
    plan, blob = compile_model(data, "silero_shaped", bind={"sr": np.array([16000], np.int64)} if args.bind_sr else None)
    t_compile = time.perf_counter() - t0
    weights = load_weights_bin(plan, blob)
    n = args.chunks * CHUNK
    rng = np.random.default_rng(0)
    t = np.arange(n) / 16000.0
    speech = (np.sin(2 * np.pi * 3 * t) > 0).astype(np.float32)           # bursts, so that the segment logic has something to do
    pcm = (speech * (0.3 * np.sin(2 * np.pi * 220 * t) + 0.05 * rng.uniform(-1, 1, n))).astype(np.float32)
    padded = np.concatenate([np.zeros(CONTEXT, np.float32), pcm])[None, :]   # [1, CONTEXT + n]
    zeros = np.zeros((1, 1, HID), np.float32)
    sr = np.array([16000], np.int64)

    class Lane:
        """one audio stream: a context (= HIP stream), the runner's workspace, the recorded per-chunk graph, the LSTM state"""

        def __init__(self, ctx):
            self.ctx = ctx
            self.runner = Runner(plan, weights, ctx)
            self.audio = TensorView(ctx.buf().upload(padded))
            self.xb = ctx.buf()
            x = self.chunk(0)
            feeds = {"x": x, "h0": TensorView(ctx.buf().upload(zeros)), "c0": TensorView(ctx.buf().upload(zeros))}
            if not args.bind_sr:
                feeds["sr"] = sr
            self.prob, hn, cn = self.runner.run(feeds)        # eager once: uploads and packs the weights, sizes every buffer
            self.calls = self.runner.calls
            # The state lives where the network writes it: the next chunk reads h0 / c0 from the buffers hn / cn were stored in
            # (the LSTM kernel takes its initial state into LDS before the first step and stores the final state after the
            # last one; tests/test_conv_rnn.py pins that in-place use), so no copy moves it between chunks.
            self.hbuf, self.cbuf = hn.raw().buf, cn.raw().buf
            self.feeds = dict(feeds, h0=TensorView(self.hbuf.upload(zeros)), c0=TensorView(self.cbuf.upload(zeros)))
            self.runner.run(self.feeds)
            ctx.sync()
            ctx.graph_begin()
            self.runner.run(self.feeds)
            self.graph = ctx.graph_end()

        def chunk(self, i):
            return K.view_copy(self.audio, [["slice", 1, i * CHUNK, CONTEXT + CHUNK]], out=self.xb, ctx=self.ctx)

        def reset(self):
            self.hbuf.upload(zeros)
            self.cbuf.upload(zeros)

        def read_prob(self):   # the device value as it is NOW (a TensorView keeps the first host copy it made): 4-byte D2H + sync
            return float(self.prob.raw().numpy().reshape(-1)[0])

    ctx = lele_amd._lib.Ctx(0)
    lane = Lane(ctx)

    def stream(read_each):
        lane.reset()
        probs = []
        for i in range(args.chunks):
            lane.chunk(i)
            lane.graph.launch()
            if read_each:
                probs.append(lane.read_prob())
        ctx.sync()
        return probs

    stream(True)
    ts, tb = [], []
    for _ in range(args.runs):
        ctx.sync()
        t0 = time.perf_counter()
        probs = stream(True)
        ts.append(time.perf_counter() - t0)
    for _ in range(args.runs):
        ctx.sync()
        t0 = time.perf_counter()
        stream(False)
        tb.append(time.perf_counter() - t0)
    # the same chunks eagerly through the runner (no graph): must give the same probabilities
    lane.reset()
    eager = []
    for i in range(args.chunks):
        lane.chunk(i)
        p = lane.runner.run(lane.feeds)[0]
        eager.append(float(p.raw().numpy().reshape(-1)[0]))
    multi = {}
    if args.streams > 1:  # independent audio streams side by side, one lane each (Silero parallelises across streams only)
        lanes = [lane] + [Lane(lele_amd._lib.Ctx(0)) for _ in range(args.streams - 1)]

        def round_():
            for ln in lanes:
                ln.reset()
            for i in range(args.chunks):
                for ln in lanes:
                    ln.chunk(i)
                    ln.graph.launch()
            for ln in lanes:
                ln.ctx.sync()
        round_()
        tm = []
        for _ in range(args.runs):
            t0 = time.perf_counter()
            round_()
            tm.append(time.perf_counter() - t0)
        same = all(ln.read_prob() == lanes[0].read_prob() for ln in lanes)
        multi = {"streams": args.streams, "streams_us_per_chunk": round(1e6 * float(np.mean(tm)) / (args.chunks * args.streams), 2),
                 "rtf_streams": round(float(np.mean(tm)) / (args.streams * n / 16000.0), 7), "streams_agree": bool(same)}
    calls = lane.calls
    seconds = n / 16000.0
    segs = apps.vad_segments(np.asarray(probs, np.float32), CHUNK, n, n)
    rec = {"config": "c1_silero_shaped", "chunks": args.chunks, "audio_s": round(seconds, 3), "kernel_calls_per_chunk": calls + 1,
           "compile_s": round(t_compile, 3), "weights_bin_bytes": len(blob), "plan_has_if": any(s["op"] == "if" for s in plan["statements"]),
           "streaming_us_per_chunk": round(1e6 * float(np.mean(ts)) / args.chunks, 2), "rtf_streaming": round(float(np.mean(ts)) / seconds, 7),
           "batched_us_per_chunk": round(1e6 * float(np.mean(tb)) / args.chunks, 2), "rtf_batched": round(float(np.mean(tb)) / seconds, 7),
           "graph_equals_eager": bool(np.array_equal(np.asarray(probs), np.asarray(eager))),
           "graph_vs_eager_max_abs": float(np.abs(np.asarray(probs) - np.asarray(eager)).max()),
           "prob_range": [round(min(probs), 6), round(max(probs), 6)], "segments": len(segs),
           "note": "assumed topology, random weights; per-chunk latency, no roofline claim"}
    rec.update(multi)
    print(json.dumps(rec), flush=True)
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        json.dump(rec, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
