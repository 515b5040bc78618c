#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: "SenseVoiceSmall steady-state RTF @16kHz, 1/2/4/8 MI355X; STFT+mel GB/s".

    python bench.py [--gpus N] [--steps K] [--warmup W]

One process per GPU.  With --gpus N > 1 and no WORLD_SIZE in the environment the script SPAWNS its N ranks itself (RANK /
LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set per child); under `python -m torch.distributed.run` it
takes the launcher's environment.  It refuses to run when the world size differs from --gpus.  Rank 0 prints ONE JSON line.

Three legs per run, all on synthetic inputs that are resident in HBM before anything is timed (1 and 2: 16 kHz audio; 3 -- `yolo` --
BASELINE configs[4]: a Yolo26n-seg-shaped network, batch 64 of 640 x 640 images per GPU as one hipGraph, 3 warm-up + 10 timed
forwards as examples/yolo26n-seg/src/benchmark.rs:29-56, priced against the f32 MFMA peak, two images checked against the batch-1
plan, the oracle's convolution route on one host core beside it):

 1. STFT+mel (BASELINE configs[1], the line's `metric` / `value`): a "step" = one pass of SenseVoiceFrontend (PCM -> log-mel ->
    LFR) over `--batch` 30 s utterances per GPU (default 2048 = 17 hours of audio, 6.2 GB of PCM and features resident in
    HBM; 256 distinct synthetic utterances repeated on the device).  EXACTLY --steps steps are timed between barrier + device
    sync fences, MAX over ranks; value = algorithmic GB/s of the whole job.  The recogniser legs run first and a step takes
    ~3.7 ms, so a run with few steps is not a measurement of the device's ~25 ms clock ramp (at 256 utterances per step the first
    40 steps read 0.72 -> 0.48 ms).  `roofline` prices the dominant kernel (fe_main_kernel), timed
    with HIP events on the stream it runs on.
 2. SenseVoice-shaped recogniser (BASELINE's headline: steady-state RTF; configs[2] and [3]) -- `sensevoice` object:
      * every N: one shard of configs[3] per GPU (32 x 10 s utterances): front-end -> CMVN -> 70-layer encoder (compiled
        plan replayed as one hipGraph) -> greedy decode on the device -> ONE RCCL all-gather of the token ids through the
        C ABI (lele_hip_comm_*).  Mean of `--sv-steps` steady-state steps, fences as above, MAX over ranks:
        `rtf_c4` = wall / audio seconds of the WHOLE job, `audio_s_per_s` its inverse (higher is better; scales with N).
      * N = 1 additionally: configs[2], one 30 s utterance, as lele's harness measures it (examples/sensevoice/src/main.rs:
        198-237: mean of 10 steady-state forwards / audio seconds): `rtf_model` (encoder graph only) and `rtf_e2e`
        (front-end + CMVN + encoder + decode + ids on the host).
    The ONNX file of SenseVoiceSmall is not in the reference tree: the topology is the ASSUMED one of SURVEY.md 8a and the
    weights are synthetic (8d) -- labelled as such in the line.  The model's dominant kernel (the quantised linear) gets its
    own roofline fields, and the oracle runs the same layer stack on one host core as the CPU baseline.

`cpu_baseline` (rank 0, N = 1 only): the CPU oracle -- a C++ restatement of lele's x86 AVX2 path, one thread = lele's
execution model -- on a bounded sample of each leg.  A reported baseline, not the target.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))

SAMPLE_RATE = 16000
SECONDS = 30
VALU_PEAK_TLANE = 78.6  # T lane-ops/s: 256 CUs x 128 lanes x 2.4 GHz
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
I8_PEAK_TOPS = 3944.0   # MI355X_MICROARCH.md: i8 MFMA >= 3944 TOPS measured (16x16x64)
F32_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: f32 vector (= f32-input MFMA) peak


def synth_batch(batch, n, seed0):
    """0.3 sin(2pi 220 t) + 0.2 sin(2pi 1000 t) + 0.05 U(-1,1), seed per utterance"""
    t = np.arange(n) / float(SAMPLE_RATE)
    tone = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
    out = np.empty((batch, n), np.float32)
    for i in range(batch):
        rng = np.random.default_rng(seed0 + i)
        out[i] = tone + (0.05 * rng.uniform(-1, 1, n)).astype(np.float32)
    return out


from lele_amd.sharded import shard_range  # noqa: E402,F401  SURVEY.md 8(e): static block partition; rank r owns utterances [lo, hi)


def rank_seed_base(rank, total, world):
    """utterance i of the global batch is synthesised from seed i, whichever rank owns it"""
    return shard_range(total, rank, world)[0]


def max_over_ranks(wall, dist, device):
    """the job's time is the slowest rank's time: one MAX all-reduce (the only collective besides the barriers)"""
    if dist is None:
        return wall
    import torch
    tt = torch.tensor([wall], dtype=torch.float64, device=device)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    return float(tt.item())


# ----------------------------------------------------------------------------------------------- CPU baselines (oracle)
def cpu_quota():
    hw = os.cpu_count() or 1
    cores = hw
    try:  # the container may be granted fewer CPUs than the host has (cgroup v2 cpu.max = "quota period")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(hw, -(-int(q) // int(per))))
    except Exception:
        pass
    return cores, hw


def cpu_baseline_frontend(n, budget_s=8.0):
    """the oracle (kind "port"), 1 thread, on utterances of the same shape until ~budget_s of CPU work"""
    from oracle import pyoracle as O
    O.lib()
    xs = synth_batch(4, n, 10_000)
    t_lfr, _ = O.frontend_shape(n)
    O.frontend_compute(xs[0])  # warm
    done, t0 = 0, time.perf_counter()
    while True:
        O.frontend_compute(xs[done % 4])
        done += 1
        el = time.perf_counter() - t0
        if el >= budget_s or done >= 4096:
            break
    bytes_per_utt = 4 * n + 4 * t_lfr * 560
    return {"value": round(done * bytes_per_utt / el / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "%d x 30 s utterances in %.1f s, oracle (C++ restatement of lele's AVX2 path), 1 thread of %d" % (done, el, os.cpu_count() or 0),
            "rtf": round(el / (done * n / SAMPLE_RATE), 6)}


def cpu_baseline_frontend_all_cores(n, budget_s=5.0):
    """lele's only route to multi-core is one independent instance per core (everything in it is
    Par::Seq with thread-local scratch).  One forked worker PROCESS per granted hardware thread, each looping over the
    oracle's front-end for `budget_s` seconds."""
    from oracle import pyoracle as O
    O.lib()
    cores, hw = cpu_quota()
    xs = synth_batch(4, n, 20_000)
    t_lfr, _ = O.frontend_shape(n)
    O.frontend_compute(xs[0])  # tables / code warm before the fork
    pipes = []
    t0 = time.perf_counter()
    for i in range(cores):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:  # child: oracle only, no HIP calls, leaves through _exit
            os.close(r)
            cnt, stop = 0, time.perf_counter() + budget_s
            while time.perf_counter() < stop:
                O.frontend_compute(xs[(i + cnt) % 4])
                cnt += 1
            os.write(w, str(cnt).encode())
            os._exit(0)
        os.close(w)
        pipes.append((pid, r))
    total = 0
    for pid, r in pipes:
        total += int(os.read(r, 64) or b"0")
        os.close(r)
        os.waitpid(pid, 0)
    el = time.perf_counter() - t0
    bytes_per_utt = 4 * n + 4 * t_lfr * 560
    return {"value": round(total * bytes_per_utt / el / 1e9, 3), "unit": "GB/s", "cores": cores, "kind": "port",
            "sample": "%d x 30 s utterances, %d single-threaded oracle processes (quota %d of %d hw threads), %.1f s" % (total, cores, cores, hw, el),
            "rtf": round(el / max(1, total) / (n / SAMPLE_RATE), 7)}


def logits_agreement(dev, ref):
    """MAE of the logits and arg-max agreement per frame, as examples/sensevoice/tests/e2e_test.rs:126-189 judges the model (its bar:
    mae <= 1.0 against ONNX Runtime), plus the MAE relative to the logits' rms"""
    dev, ref = np.asarray(dev, np.float64), np.asarray(ref, np.float64)
    rms = float(np.sqrt(np.mean(ref * ref)))
    return {"frames": int(ref.shape[0] * ref.shape[1]), "argmax_agreement": round(float((dev.argmax(-1) == ref.argmax(-1)).mean()), 4),
            "mae": round(float(np.abs(dev - ref).mean()), 4), "mae_over_rms": round(float(np.abs(dev - ref).mean()) / rms, 5)}


def cpu_baseline_model(enc_arrays, feats, budget_s=14.0, dev_logits=None):
    """lele's execution model for configs[2] on ONE host core: the oracle's restatement of every kernel the generated code
    would call (oracle/sensevoice_ref.py), layer after layer on one 30 s utterance, until ~budget_s of CPU work; the
    remaining layers are extrapolated from the per-layer mean (layers 1..69 are identical in shape), the CTC head is timed
    once.  Integer GEMM = lele's vpmaddwd scheme; the two f32 attention products use the oracle's own AVX2-FMA kernel where
    lele calls the faer crate (not in the tree)."""
    from oracle import sensevoice_ref as R
    from oracle import pyoracle as O
    x = np.concatenate([enc_arrays["prompt"], feats], axis=1).astype(np.float32)
    per, t_all = [], time.perf_counter()
    for L in enc_arrays["layers"]:
        t0 = time.perf_counter()
        x = R.layer_forward(x, L)
        per.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > budget_s:
            break
    t0 = time.perf_counter()
    xn = O.layer_norm(x, enc_arrays["ln_out"][0], enc_arrays["ln_out"][1], -1, 1e-5)
    ref_logits = R.qlinear(xn, enc_arrays["ctc"])
    head = time.perf_counter() - t0
    nl = len(enc_arrays["layers"])
    rest = per[1:] if len(per) > 1 else per
    est = sum(per) + (nl - len(per)) * (sum(rest) / len(rest)) + head
    audio = feats.shape[0] * SECONDS
    out = {"model_rtf": round(est / audio, 6), "model_ms": round(est * 1e3, 1), "model_cores": 1,
           "model_sample": "%d of %d layers + CTC head of ONE 30 s utterance timed (%.1f s), rest extrapolated per layer; oracle, 1 thread"
                           % (len(per), nl, sum(per) + head)}
    if dev_logits and len(per) == nl:   # the oracle ran the whole forward: the checker's logits against the device's (the oracle as CHECKER)
        out["oracle_agreement"] = {"what": "configs[2] logits, device vs the oracle's forward timed here (e2e_test.rs:126-189: MAE, arg-max per frame; "
                                           "its MAE bar is 1.0); bars asserted in tests/test_graph_oracle.py",
                                   **{k: logits_agreement(v, ref_logits) for k, v in dev_logits.items()}}
    return out


def cpu_baseline_model_all_cores(enc_arrays, feats, budget_s=6.0):
    """BASELINE.md section 3 `cpu-Nt` for the recogniser: lele runs one forward on one thread (Par::Seq everywhere), so its only route
    to many cores is one independent instance per core.  One forked worker per granted hardware thread, each running the oracle's
    layer stack (oracle/sensevoice_ref.py) on its own copy of one 30 s utterance for `budget_s` seconds; throughput = layers finished
    over all workers, scaled to whole 70-layer forwards (+ the CTC head's share as timed on one core)."""
    from oracle import sensevoice_ref as R
    cores, hw = cpu_quota()
    x0 = np.concatenate([enc_arrays["prompt"], feats], axis=1).astype(np.float32)
    layers = enc_arrays["layers"]
    R.layer_forward(x0, layers[0])  # code / tables warm before the fork
    pipes = []
    t0 = time.perf_counter()
    for i in range(cores):
        r, w = os.pipe()
        pid = os.fork()
        if pid == 0:  # child: oracle only, no HIP calls, leaves through _exit
            os.close(r)
            cnt, stop = 0, time.perf_counter() + budget_s
            x = R.layer_forward(x0, layers[0])
            while time.perf_counter() < stop:
                x = R.layer_forward(x, layers[1 + cnt % (len(layers) - 1)])
                cnt += 1
            os.write(w, str(cnt).encode())
            os._exit(0)
        os.close(w)
        pipes.append((pid, r))
    total = 0
    for pid, r in pipes:
        total += int(os.read(r, 64) or b"0")
        os.close(r)
        os.waitpid(pid, 0)
    el = time.perf_counter() - t0
    nl = len(layers)
    fwd_per_s = total / float(nl) / el if el > 0 else 0.0
    audio = feats.shape[0] * SECONDS
    return {"model_rtf_all_cores": round(1.0 / (fwd_per_s * audio), 7) if fwd_per_s > 0 else None, "model_all_cores": cores,
            "model_all_cores_sample": "%d layer forwards of a 30 s utterance by %d single-threaded oracle processes (quota %d of %d hw threads) in %.1f s "
                                      "= %.2f 70-layer forwards per second (the CTC head, ~3 %% of a forward, not included)" % (total, cores, cores, hw, el, fwd_per_s)}


# ----------------------------------------------------------------------------------------------- the two legs
def frontend_leg(args, ctx, rank, world, fence, dist, device):
    from lele_amd.features import SenseVoiceFrontend
    fe = SenseVoiceFrontend(ctx=ctx)
    n = SAMPLE_RATE * SECONDS
    t_lfr, cols, nf = fe.out_rows(n)
    bytes_per_utt = 4 * n + 4 * t_lfr * cols  # PCM read once + LFR written once
    # weak scaling: the global batch is world * batch utterances; this rank synthesises and keeps its own shard
    lo, hi = shard_range(world * args.batch, rank, world)
    # resident in HBM before the timed region: up to DISTINCT utterances are synthesised on the host (seeded per global index),
    # a larger batch repeats them on the device -- the kernel's work per utterance does not depend on the samples
    from lele_amd import kernels as K
    DISTINCT = 256
    mine = hi - lo
    base = ctx.buf().upload(synth_batch(min(mine, DISTINCT), n, rank_seed_base(rank, world * args.batch, world)))
    if mine > DISTINCT:
        reps = -(-mine // DISTINCT)
        pbuf = ctx.buf()
        pcm = K.tile(base, [reps, 1], out=pbuf, ctx=ctx)
        if reps * DISTINCT != mine:
            pbuf2 = ctx.buf()
            pcm = K.slice(pcm, [0], [mine], [0], out=pbuf2, ctx=ctx)
            ctx.sync()
            pbuf.close()
            pbuf = pbuf2
        ctx.sync()
        base.buf.close()
    else:
        pcm, pbuf = base, base.buf
    out = ctx.buf()
    for _ in range(args.warmup):
        fe.compute_batch(pcm, out)
    fence()
    fe.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.compute_batch(pcm, out)
    fence()
    wall = time.perf_counter() - t0
    sum_ms, main_ms, runs = fe.profile_read()
    fe.set_profiling(False)
    wall = max_over_ranks(wall, dist, device)
    pbuf.close()
    out.close()
    return {"wall": wall, "n": n, "t_lfr": t_lfr, "cols": cols, "nf": nf, "bytes_per_utt": bytes_per_utt, "main_ms": main_ms,
            "sum_ms": sum_ms, "runs": runs}


def open_comm(args, ctx, rank, world, dist, device):
    """N > 1: the C ABI's own RCCL communicator (the 128-byte id travels through the already-initialised process group), shared by the
    recogniser's id gather and the Yolo leg's detection gather.  -> (comm or None, note when torch.distributed stands in, ranks the
    communicator itself reached)"""
    import lele_amd
    if world <= 1:
        return None, None, None
    comm, comm_note, seen = None, None, None
    uid, err = [None], ""
    try:
        if rank == 0:
            uid = [lele_amd._lib.Comm.unique_id()]
    except Exception as e:  # noqa: BLE001
        err = str(e)
    dist.broadcast_object_list(uid, src=0)
    if uid[0] is not None:
        try:
            comm = lele_amd._lib.Comm.from_id(ctx, uid[0], rank, world)
            probe = comm.allreduce_max(rank)
            if probe != world - 1:
                raise RuntimeError("communicator probe returned %r" % probe)
        except Exception as e:  # noqa: BLE001
            err, comm = str(e), None
    # every rank must take the same route: agree (MIN over ranks of "mine works"); otherwise the results travel through the process
    # group torch already holds, and the line says so -- a scaling measurement is not lost to a transport problem
    import torch
    ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=device)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()) == 0:
        if not args.allow_fallback:  # a scaling number measured on the wrong transport is worse than none
            raise SystemExit("bench.py: the C ABI's RCCL communicator was not usable on every rank (%s); rerun with --allow-fallback to "
                             "measure with torch.distributed's all_gather instead" % (err or "another rank failed"))
        if comm is not None:
            comm.close()
        comm, comm_note = None, "torch.distributed all_gather (the C ABI communicator was not usable on every rank: %s)" % (err or "another rank failed")
    elif comm is not None:  # how many ranks the communicator itself reached (a MAX all-reduce of rank + 1 through the C ABI)
        seen = int(comm.allreduce_max(rank + 1))
    return comm, comm_note, seen


def sensevoice_leg(args, ctx, rank, world, fence, dist, device, comm=None, comm_note=None, ranks_seen=None):
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd.compiler import compile_model
    from lele_amd.features import Cmvn, SenseVoiceFrontend
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.sharded import all_gather_ids, all_gather_ids_rccl
    from sensevoice_graph import VOCAB, Encoder, encoder_arrays, encoder_onnx

    rec = {"topology": "ASSUMED (SURVEY.md 8a): 70 SAN-M layers, d=512, 4x128 heads, FFN 2048, FSMN k=11, int8 linears, CTC 25055; "
                       "synthetic weights (8d)", "layers": args.layers,
           # parity bars of what runs here (tests/): everything quantised is bit-exact against the oracle; the fused attention kernel
           # (split-bf16 products, v_exp_f32) is held to 2e-4 element-relative against the oracle's three-operator composition, and at
           # full size to <= 1e-5 of the elements beyond that, none beyond 1e-2 of the rms (tests/test_attention.py,
           # tests/test_fullsize_graph.py); LELE_HIP_ATTENTION_EXACT=1 selects the bit-exact replica of the three calls
           "parity": {"quantised_linears": "bit-exact", "attention_bar": 2e-4, "attention_outliers_allowed": 1e-5,
                      "attention_outlier_ceiling_of_rms": 1e-2, "other_f32": 1e-4}}
    fe, cmvn = SenseVoiceFrontend(ctx=ctx), Cmvn(ctx=ctx)
    # damped residual branches (tools/sensevoice_graph.py): kernel times do not depend on the values, and the whole forward can then be held
    # against the oracle's (`oracle_agreement` below; tests/test_graph_oracle.py)
    enc = Encoder(ctx, args.layers, damped=True)
    rec["topology"] += "; residual branches damped (DAMP_BRANCH %.3g, DAMP_V %.3g, DAMP_FIRST %.3g)" % tuple(
        __import__("sensevoice_graph").__dict__[k] for k in ("DAMP_BRANCH", "DAMP_V", "DAMP_FIRST"))
    skip = np.zeros(VOCAB, np.uint8)          # blank + a block of <|...|> specials, as tokenizer.rs:38-48 marks them
    skip[0] = 1
    skip[VOCAB - 200:] = 1
    skip = lele_amd._lib.Weight(skip)

    if ranks_seen is not None:
        rec["rccl_ranks_seen"] = ranks_seen

    def build(batch, seconds, seed0):
        n = SAMPLE_RATE * seconds
        pcm = ctx.buf().upload(synth_batch(batch, n, seed0))
        plan, blob = compile_model(encoder_onnx(enc, batch), "sensevoice_shaped")
        runner = Runner(plan, load_weights_bin(plan, blob), ctx)
        fbuf, cbuf, abuf, ibuf, nbuf = (ctx.buf() for _ in range(5))

        def features():
            f = fe.compute_batch(pcm, fbuf)                                    # [B, T, 560]
            return cmvn.compute(f, out=cbuf)

        feats = features()
        runner.run({"feats": feats})          # eager once: uploads and packs every weight, sizes every buffer
        ctx.sync()
        ctx.graph_begin()
        logits = runner.run({"feats": feats})[0]
        graph = ctx.graph_end()
        graph.launch()
        ctx.sync()

        def decode():
            return K.token_filter(K.argmax_last(logits, out=abuf, ctx=ctx), skip, out_ids=ibuf, out_counts=nbuf, ctx=ctx)

        # the whole step as ONE graph: front-end + CMVN, the encoder plan, arg-max + token filter (same kernels, same buffers as the three
        # pieces above; what stays outside is the exchange of the ids).  Falls back to the pieces if the capture is refused.
        step_graph, step_out = None, None
        try:
            decode()
            ctx.sync()
            ctx.graph_begin()
            f2 = features()
            runner.run({"feats": f2})
            step_out = decode()
            step_graph = ctx.graph_end()
            step_graph.launch()
            ctx.sync()
        except Exception:  # noqa: BLE001
            try:
                ctx.graph_abort()
            except Exception:  # noqa: BLE001
                pass
            step_graph = None
        return {"features": features, "graph": graph, "decode": decode, "feats": feats, "logits": logits, "runner": runner,
                "plan": plan, "audio": batch * seconds, "step_graph": step_graph, "step_out": step_out}

    # ---- configs[3] shard: per_gpu x 10 s utterances on every rank, ids gathered over RCCL
    total = args.per_gpu * world
    lo, hi = shard_range(total, rank, world)
    c4 = build(hi - lo, 10, lo)
    gbufs = [ctx.buf() for _ in range(4)]

    def step_c4():
        if c4["step_graph"] is not None:       # PCM -> features -> encoder -> token ids as one recorded graph
            c4["step_graph"].launch()
            ids, counts = c4["step_out"]
        else:
            c4["features"]()                   # same buffers every step: the graph reads the CMVN output buffer
            c4["graph"].launch()
            ids, counts = c4["decode"]()
        if comm is not None:
            return all_gather_ids_rccl(ids, counts, total, comm, ctx, gbufs)
        if world > 1:
            return all_gather_ids(ids.numpy(), counts.numpy(), total, dist, device)
        return all_gather_ids(ids.numpy(), counts.numpy(), total)       # single process: just the D2H copy of the ids

    for _ in range(2):
        everything = step_c4()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.sv_steps):
        everything = step_c4()
    fence()
    wall = max_over_ranks(time.perf_counter() - t0, dist, device)
    mine = all_gather_ids(*[a.numpy() for a in c4["decode"]()], hi - lo)
    agree = all(np.array_equal(a, b) for a, b in zip(everything[lo:hi], mine)) and len(everything) == total
    fn_count = {}
    for st in c4["plan"]["statements"]:
        if st["op"] == "call":
            fn_count[st["fn"]] = fn_count.get(st["fn"], 0) + 1
    rec.update({"c4_utterances": total, "c4_utterances_per_gpu": args.per_gpu, "c4_seconds_per_utterance": 10,
                "c4_ms_per_step": round(1e3 * wall / args.sv_steps, 3), "rtf_c4": round(wall / args.sv_steps / (total * 10), 8),
                "audio_s_per_s": round(total * 10 * args.sv_steps / wall, 1), "c4_steps": args.sv_steps,
                "c4_collective": "rccl all-gather of token ids via lele_hip_comm_allgather_i32" if comm else (comm_note or "none (single process)"),
                "c4_gathered_ok": bool(agree), "c4_tokens": int(c4["logits"].shape[1]),
                "c4_step_as_one_graph": c4["step_graph"] is not None,
                "plan_statements": len(c4["plan"]["statements"]), "plan_calls": sum(fn_count.values()),
                "logits_finite": bool(np.isfinite(c4["logits"].numpy()[0]).all())})

    # ---- section 8(e), "full logits if requested": the raw [T + 4, 25055] tensors of every utterance on every rank (--gather-logits)
    if args.gather_logits:
        from lele_amd.sharded import all_gather_logits, all_gather_logits_rccl
        lbufs = [ctx.buf() for _ in range(2)]

        def gather_logits():
            if comm is not None:
                return all_gather_logits_rccl(c4["logits"], total, comm, ctx, lbufs)
            if world > 1:
                return all_gather_logits(c4["logits"].numpy(), total, dist, device)
            return all_gather_logits(c4["logits"].numpy(), total)
        every = gather_logits()
        fence()
        t0 = time.perf_counter()
        every = gather_logits()
        fence()
        lwall = max_over_ranks(time.perf_counter() - t0, dist, device)
        rec["c4_logits_gather"] = {"what": "f32 logits [utterances, T + 4, 25055] of the whole batch on every rank (SURVEY.md 8(e): 17.1 MB an utterance at "
                                           "10 s), incl. the read-back to the host", "utterances": int(every.shape[0]), "tokens": int(every.shape[1]),
                                   "bytes_per_rank": int((hi - lo) * every.shape[1] * every.shape[2] * 4), "ms": round(1e3 * lwall, 3),
                                   "collective": "rccl all-gather via lele_hip_comm_allgather" if comm else (comm_note or "none (single process)"),
                                   "own_block_equals_local_logits": bool(np.array_equal(every[lo:hi], c4["logits"].numpy()))}

    # ---- the quantised linear (the model's dominant kernel) at the configs[3] shard shape, per-stage HIP events
    if rank == 0:
        L = enc.layers[1]
        x = ctx.buf().upload(np.random.default_rng(5).standard_normal((hi - lo, c4["logits"].shape[1], 512)).astype(np.float32))
        xn = K.layer_norm(x, L.ln1[0], L.ln1[1], -1, 1e-5, out=ctx.buf(), ctx=ctx)
        ob = ctx.buf()
        p = L.ffn1
        for _ in range(5):
            K.fused_quantized_linear(xn, p.w, p.scale, p.zero, p.bias, True, out=ob, ctx=ctx)
        ctx.quant_set_profiling(True)
        for _ in range(50):
            K.fused_quantized_linear(xn, p.w, p.scale, p.zero, p.bias, True, out=ob, ctx=ctx)
        r_ms, q_ms, g_ms, calls = ctx.quant_profile_read()
        ctx.quant_set_profiling(False)
        # the op as it runs in the model: 20 calls recorded into a hipGraph, the replays timed with HIP events on the ctx stream (the
        # per-stage events above put a host-visible boundary between the stages; their sum is the larger, staged figure)
        ctx.sync()
        ctx.graph_begin()
        for _ in range(20):
            K.fused_quantized_linear(xn, p.w, p.scale, p.zero, p.bias, True, out=ob, ctx=ctx)
        gq = ctx.graph_end()
        gq.launch()
        ctx.sync()
        ctx.timer_start()
        for _ in range(10):
            gq.launch()
        op_ms = ctx.timer_stop() / 200.0
        gq.close()
        m_rows, kk, nn = (hi - lo) * c4["logits"].shape[1], 512, 2048
        byts = 4 * m_rows * kk + kk * nn + 8 * nn + 4 * m_rows * nn     # f32 in, u8 weights, scale+bias, f32 out
        ops = 2 * m_rows * kk * nn
        rec["qlinear"] = {"shape": "[%d x %d] x [%d x %d] (ffn1 of one configs[3] shard, input = LayerNorm output)" % (m_rows, kk, kk, nn),
                          "range_ms": round(r_ms, 5), "quantise_ms": round(q_ms, 5), "gemm_ms": round(g_ms, 5),
                          "staged_sum_ms": round(r_ms + q_ms + g_ms, 5), "op_ms": round(op_ms, 5),
                          "calls": calls, "algorithmic_bytes": byts, "int_ops": ops,
                          "stages": "range = the {min, max} pairs come with the LayerNorm output: nothing is launched; quantise = nothing is launched either "
                                    "since round 5 (the GEMM's loader waves reduce the parameters and quantise the f32 rows: igemm_rs.h, FQ) -- "
                                    "both intervals are the cost of the two HIP events around them; gemm = igemm_rs_kernel",
                          "hbm_gbs": round(byts / (op_ms * 1e-3) / 1e9, 1) if op_ms > 0 else None,
                          "tops": round(ops / (op_ms * 1e-3) / 1e12, 1) if op_ms > 0 else None}

        # the layer's fused feed-forward block as the model runs it: first product (range + quantise in one launch), second product +
        # residual Add + the next LayerNorm = two launches; 20 calls recorded into a hipGraph, replays timed with HIP events
        try:
            L2 = enc.layers[2]
            x1 = ctx.buf().upload(np.random.default_rng(6).standard_normal((hi - lo, c4["logits"].shape[1], 512)).astype(np.float32))
            x1n = K.layer_norm(x1, L.ln2[0], L.ln2[1], -1, 1e-5, out=ctx.buf(), ctx=ctx)
            o2 = [ctx.buf(), ctx.buf()]
            f1, f2 = L.ffn1, L.ffn2

            def block():
                return K.fused_ffn_quantized_ln(x1n, f1.w, f1.scale, f1.zero, f1.bias, f2.w, f2.scale, f2.zero, f2.bias, False, x1, None,
                                                L2.ln1[0], L2.ln1[1], 1e-5, outs=o2, ctx=ctx)
            for _ in range(3):
                block()
            ctx.sync()
            ctx.graph_begin()
            for _ in range(20):
                block()
            gb = ctx.graph_end()
            gb.launch()
            ctx.sync()
            ctx.timer_start()
            for _ in range(10):
                gb.launch()
            blk_ms = ctx.timer_stop() / 200.0
            gb.close()
            m_rows = (hi - lo) * c4["logits"].shape[1]
            bbytes = 4 * m_rows * 512 * 4 + 512 * 2048 * 2 + 8 * (2048 + 512) + 8 * 512   # x1n, x1 in; sum and its LayerNorm out; both weights; scales, biases, LN
            bops = 2 * 2 * m_rows * 512 * 2048                                             # the two products (the first one's recompute is not counted)
            rec["ffn_block"] = {"what": "fused_ffn_quantized_ln [%d x 512] -> 2048 -> 512 + residual + LayerNorm: igemm_rs_kernel<3> + igemm_ask_kernel" % m_rows,
                                "op_ms": round(blk_ms, 5), "algorithmic_bytes": bbytes, "int_ops": bops,
                                "hbm_gbs": round(bbytes / (blk_ms * 1e-3) / 1e9, 1), "tops": round(bops / (blk_ms * 1e-3) / 1e12, 1)}
        except Exception as e:  # noqa: BLE001
            rec["ffn_block"] = {"failed": str(e)}

    # ---- configs[2]: one 30 s utterance, lele's own protocol (N = 1 only)
    if world == 1:
        c3 = build(1, 30, 0)
        runs = max(10, args.sv_steps)
        t_model, t_e2e = [], []
        for _ in range(runs):
            ctx.sync()
            t0 = time.perf_counter()
            c3["graph"].launch()
            ctx.sync()
            t_model.append(time.perf_counter() - t0)
        for _ in range(runs):
            ctx.sync()
            t0 = time.perf_counter()
            c3["features"]()
            c3["graph"].launch()
            ids, counts = c3["decode"]()
            ids.numpy(), counts.numpy()        # the transcript's ids on the host: waits for the stream
            t_e2e.append(time.perf_counter() - t0)
        rec.update({"c3_tokens": int(c3["logits"].shape[1]), "c3_runs": runs,
                    "c3_model_ms": round(1e3 * float(np.mean(t_model)), 3), "c3_e2e_ms": round(1e3 * float(np.mean(t_e2e)), 3),
                    "rtf_model": round(float(np.mean(t_model)) / 30.0, 7), "rtf_e2e": round(float(np.mean(t_e2e)) / 30.0, 7),
                    "rtf_target": 0.001})
        rec["_c3_feats"] = c3["feats"].numpy()
        rec["_enc"] = enc
        rec["_c3_logits"] = {"shipped": c3["logits"].numpy().copy()}
        # the same two figures with the attention on its bit-exact replica path (LELE_HIP_ATTENTION_EXACT=1: f32 MFMA in the tiled GEMM's
        # k order + the reference's row softmax = the bits of the three-call sequence): graphs re-recorded under the switch
        old = os.environ.get("LELE_HIP_ATTENTION_EXACT")
        os.environ["LELE_HIP_ATTENTION_EXACT"] = "1"
        try:
            c3x = build(1, 30, 0)
            tx = []
            for _ in range(runs):
                ctx.sync()
                t0 = time.perf_counter()
                c3x["graph"].launch()
                ctx.sync()
                tx.append(time.perf_counter() - t0)
            c4x = build(hi - lo, 10, lo)

            def step_c4x():
                c4x["features"]()
                c4x["graph"].launch()
                ids, counts = c4x["decode"]()
                return all_gather_ids(ids.numpy(), counts.numpy(), total)
            for _ in range(2):
                ex_ids = step_c4x()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(args.sv_steps):
                ex_ids = step_c4x()
            ctx.sync()
            wx = time.perf_counter() - t0
            rec["_c3_logits"]["exact"] = c3x["logits"].numpy().copy()
            rec.update({"rtf_model_exact": round(float(np.mean(tx)) / 30.0, 7), "c3_model_ms_exact": round(1e3 * float(np.mean(tx)), 3),
                        "rtf_c4_exact": round(wx / args.sv_steps / (total * 10), 8), "c4_ms_per_step_exact": round(1e3 * wx / args.sv_steps, 3),
                        "exact_vs_shipped_c4": logits_agreement(c4["logits"].numpy(), c4x["logits"].numpy()),
                        "exact_note": "LELE_HIP_ATTENTION_EXACT=1: attention = the bits of matmul -> softmax -> matmul.  `exact_vs_shipped_c4`: the two "
                                      "device paths against each other (1e-6 apart at every attention output, bit-identical elsewhere) -- the "
                                      "sensitivity of a 70-layer stack of DYNAMIC u8 quantisers to a last-bit difference, i.e. the floor of any "
                                      "end-to-end comparison; `oracle_agreement` holds each path against the oracle's forward of configs[2]"})
        finally:
            if old is None:
                os.environ.pop("LELE_HIP_ATTENTION_EXACT", None)
            else:
                os.environ["LELE_HIP_ATTENTION_EXACT"] = old
    return rec


def yolo_conv_layers(plan, shapes):
    """the 2-D convolutions of a compiled plan with their run-time shapes: (fn, x shape, w shape, dilations, group, pads, strides)"""
    def ints(node):
        return [int(v["int"]) for v in node.get("list", [])]
    layers = []
    for st in plan["statements"]:
        if st.get("op") == "call" and st.get("fn") in ("conv2d", "conv2d_silu", "conv2d_fused") and "ref" in st["args"][0]:
            a = st["args"]
            layers.append((st["fn"], shapes[a[0]["ref"]], a[1]["weight"][3], ints(a[3]), int(a[4]["int"]), ints(a[5]), ints(a[6])))
    return layers


def cpu_baseline_yolo(layers):
    """lele's Conv2d route on ONE host core for ONE image: the oracle's im2col + GEMM + bias / activation pass (oracle/conv_fast.cpp,
    conv2d.rs:597-760; the product is the oracle's own AVX2-FMA kernel where lele calls faer) over every convolution of the network,
    on random operands of the layers' shapes."""
    from oracle import pyoracle as O
    O.lib()
    rng = np.random.default_rng(7)
    t_all, macs = 0.0, 0
    for fn, xs, ws, dil, group, pads, strides in layers:
        x = rng.standard_normal([1] + list(xs[1:])).astype(np.float32)
        w = rng.standard_normal(ws).astype(np.float32)
        b = rng.standard_normal(ws[0]).astype(np.float32)
        t0 = time.perf_counter()
        y = O.conv2d_im2col(x, w, b, dil, group, pads, strides, "silu" if fn == "conv2d_silu" else None)
        t_all += time.perf_counter() - t0
        macs += y.size * ws[1] * ws[2] * ws[3]
    return {"value": round(1.0 / t_all, 3), "unit": "images/s", "cores": 1, "kind": "port", "ms_per_image": round(1e3 * t_all, 1),
            "gflops": round(2 * macs / t_all / 1e9, 2),
            "sample": "the %d convolutions of ONE 640 x 640 image (%.2f GFLOP) in %.2f s: oracle/conv_fast.cpp (im2col + the oracle's own "
                      "AVX2-FMA GEMM where lele calls faer + bias / SiLU pass), 1 thread" % (len(layers), 2 * macs / 1e9, t_all)}


def _silero_chain_weights():
    rng = np.random.default_rng(11)
    w = {"stft": (rng.standard_normal((258, 1, 256)) / 16).astype(np.float32)}
    for i, (ci, co) in enumerate([(129, 128), (128, 64), (64, 64), (64, 128)]):
        w["c%d" % i] = (rng.standard_normal((co, ci, 3)) / np.sqrt(3 * ci)).astype(np.float32)
        w["b%d" % i] = (rng.standard_normal(co) * 0.05).astype(np.float32)
    w["lw"] = (rng.standard_normal((1, 512, 128)) / np.sqrt(128)).astype(np.float32)
    w["lr"] = (rng.standard_normal((1, 512, 128)) / np.sqrt(128)).astype(np.float32)
    w["lb"] = (rng.standard_normal((1, 1024)) * 0.05).astype(np.float32)
    w["out"] = (rng.standard_normal((128, 1)) / np.sqrt(128)).astype(np.float32)
    return w


def _zh_wav_chunks():
    """the reference's fixtures/zh.wav (tests/golden/zh.wav: 89 472 samples, s16 mono 16 kHz) as 175 chunks of 512 samples (BASELINE
    configs[0]; examples/silero/src/main.rs:151-228 pads the tail with zeros)"""
    path = os.path.join(ROOT, "tests", "golden", "zh.wav")
    if not os.path.exists(path):
        return None
    b = open(path, "rb").read()
    payload = b[b.index(b"data") + 8:]
    pcm = np.frombuffer(payload[:len(payload) // 2 * 2], "<i2").astype(np.float32) / np.float32(32768.0)
    pcm = np.concatenate([pcm, np.zeros(175 * 512 - pcm.size, np.float32)])
    return pcm.reshape(175, 512)


def configs0_leg(ctx):
    """BASELINE configs[0] ("Silero VAD on fixtures/zh.wav ... plumbing, no GPU" upstream) as the DEVICE runs it: a Silero-shaped streaming
    chain (assumed topology -- the model file is not in the reference tree -- STFT as a strided conv1d, magnitude, four k = 3 conv
    blocks, one LSTM step with the state carried on the device, 1 x 1 head + sigmoid; synthetic weights; the chain
    tests/test_real_audio.py holds against the oracle at 1e-4 over all 175 chunks), one hipGraph replay per 32 ms chunk of the real
    recording, the speech probability read on the host after every chunk as examples/silero/src/main.rs:151-228 does."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    chunks = _zh_wav_chunks()
    if chunks is None:
        return None
    w = {k: Weight(v) for k, v in _silero_chain_weights().items()}
    b = [ctx.buf() for _ in range(24)]
    xb, hb, cb = ctx.buf(), ctx.buf(), ctx.buf()
    zeros = np.zeros((1, 1, 128), np.float32)

    def step():
        from lele_amd.tensor import TensorView
        from lele_amd._lib import DevTensor
        x = TensorView(DevTensor(xb, (1, 1, 512), np.float32))
        h0, c0 = TensorView(DevTensor(hb, (1, 1, 128), np.float32)), TensorView(DevTensor(cb, (1, 1, 128), np.float32))
        xp = K.pad(x, [0, 0, 64, 0, 0, 64], None, "reflect", out=b[0], ctx=ctx)
        s = K.conv1d(xp, w["stft"], None, [1], 1, [0, 0], [128], out=b[1], ctx=ctx)
        re, im = K.slice(s, [0], [129], [1], [1], out=b[2], ctx=ctx), K.slice(s, [129], [258], [1], [1], out=b[3], ctx=ctx)
        mag = K.sqrt(K.add(K.mul(re, re, out=b[4], ctx=ctx), K.mul(im, im, out=b[5], ctx=ctx), out=b[6], ctx=ctx), out=b[7], ctx=ctx)
        y = mag
        for i, st in enumerate((1, 2, 2, 1)):
            y = K.conv1d_fused(y, w["c%d" % i], w["b%d" % i], [1], 1, [1, 1], [st], True, out=b[8 + i], ctx=ctx)
        feat = K.reduce_mean(y, [2], False, out=b[12], ctx=ctx)
        _yy, h2, _c2 = K.lstm(K.reshape(feat, [1, 1, 128]), w["lw"], w["lr"], w["lb"], None, h0, c0, outs=[b[13], hb, cb], ctx=ctx)  # state in place
        return K.sigmoid(K.matmul(K.reshape(h2, [1, 128]), w["out"], out=b[14], ctx=ctx), out=b[15], ctx=ctx)
    hb.upload(zeros)
    cb.upload(zeros)
    xb.upload(chunks[0].reshape(1, 1, 512))
    step()
    ctx.sync()
    ctx.graph_begin()
    prob = step()
    graph = ctx.graph_end()

    def stream():
        hb.upload(zeros)
        cb.upload(zeros)
        out = []
        for c in chunks:
            xb.upload(c.reshape(1, 1, 512))
            graph.launch()
            out.append(float(prob.raw().numpy().reshape(-1)[0]))
        return out
    stream()
    t0 = time.perf_counter()
    runs = 3
    for _ in range(runs):
        probs = stream()
    dt = (time.perf_counter() - t0) / runs
    graph.close()
    return {"workload": "BASELINE configs[0]: Silero-SHAPED streaming VAD chain (assumed topology, synthetic weights) over the reference's fixtures/zh.wav, "
                        "175 chunks of 512 samples, LSTM state on the device, probability read on the host after every chunk",
            "chunks": 175, "audio_s": 5.6, "kernels_per_chunk": 17, "device_us_per_chunk": round(1e6 * dt / 175, 2), "rtf": round(dt / 5.6, 7),
            "prob_range": [round(min(probs), 6), round(max(probs), 6)], "_probs": probs,
            "note": "latency-bound by construction (B = 1 recurrences, one chunk every 32 ms): per-chunk latency, no roofline claim"}


def cpu_baseline_configs0(dev_probs):
    """the same chain on the oracle, one thread, all 175 chunks, in BOTH of its build modes: `scalar` -- the restatement of the
    `cfg(not(any(x86_64, aarch64, wasm32)))` bodies configs[0] names (oracle/scalar.cpp: one-channel conv1d as running sums, the LSTM's
    gate stage and the sigmoid through libm) -- and the x86 AVX2 bodies; the probabilities of each against the device's"""
    from oracle import npref
    from oracle import pyoracle as O
    chunks = _zh_wav_chunks()
    w = _silero_chain_weights()

    def chain():
        h = c = np.zeros((1, 1, 128), np.float32)
        probs = []
        t0 = time.perf_counter()
        for ch in chunks:
            x = npref.pad(ch.reshape(1, 1, 512), [0, 0, 64, 0, 0, 64], 0.0, "reflect")
            s = O.conv1d(x, w["stft"], None, [1], 1, [0, 0], [128])
            re, im = npref.slice_(s, [0], [129], [1], [1]), npref.slice_(s, [129], [258], [1], [1])
            y = np.sqrt(re * re + im * im)
            for i, st in enumerate((1, 2, 2, 1)):
                y = O.conv1d(y, w["c%d" % i], w["b%d" % i], [1], 1, [1, 1], [st], True)
            feat = npref.reduce("mean", y, [2], False)
            _yy, h, c = O.lstm(feat.reshape(1, 1, 128), w["lw"], w["lr"], w["lb"], h, c)
            probs.append(float(O.unary("sigmoid", O.matmul(h.reshape(1, 128), w["out"])).reshape(-1)[0]))
        return time.perf_counter() - t0, np.asarray(probs)
    with O.scalar():
        dt_s, p_s = chain()
    dt_a, p_a = chain()
    dev = np.asarray(dev_probs)
    return {"value": round(1e3 * dt_s / 175, 4), "unit": "ms per 32 ms chunk", "cores": 1, "kind": "port", "mode": "scalar", "rtf": round(dt_s / 5.6, 6),
            "sample": "all 175 chunks of zh.wav through the oracle in its SCALAR mode (the cfg(not(x86_64 ...)) bodies configs[0] names) in %.2f s, "
                      "and again through its restatement of lele's x86 AVX2 kernels in %.2f s (`avx2_ms_per_chunk`)" % (dt_s, dt_a),
            "avx2_ms_per_chunk": round(1e3 * dt_a / 175, 4), "avx2_rtf": round(dt_a / 5.6, 6),
            "max_abs_diff_device_vs_oracle_probability": round(float(np.abs(p_a - dev).max()), 9),
            "max_abs_diff_device_vs_scalar_oracle_probability": round(float(np.abs(p_s - dev).max()), 9),
            "max_abs_diff_scalar_vs_avx2_oracle_probability": round(float(np.abs(p_s - p_a).max()), 9)}


def cpu_baseline_yolo_graph(lifted):
    """configs[4]'s CPU baseline on the reference's own graph: ONE forward of the lifted Yolo26n-seg call sequence, statement by
    statement on the oracle (oracle/plan_ref.py: im2col + the oracle's AVX2-FMA GEMM where lele calls faer, AVX2 epilogues, the index
    operators in numpy), one thread -- lele's execution model (Par::Seq, one image a call)."""
    import lift_generated as L
    from oracle import plan_ref
    plan = json.load(open(lifted))
    raw = L.synth_weights(plan, dict(L.DEFAULT_CONSTS))
    x = np.random.default_rng(1000).uniform(0, 1, (1, 3, 640, 640)).astype(np.float32)
    ref = plan_ref.PlanRef(plan, raw)
    ref.run({plan["inputs"][-1]: x})
    t0 = time.perf_counter()
    runs = 3
    for _ in range(runs):
        ref.run({plan["inputs"][-1]: x})
    dt = (time.perf_counter() - t0) / runs
    return {"value": round(1.0 / dt, 3), "unit": "images/s", "cores": 1, "kind": "port", "ms_per_image": round(1e3 * dt, 1),
            "gflops": round(9.127 / dt, 2),
            "sample": "%d whole forwards of ONE 640 x 640 image through the reference's generated graph (%d kernel statements, 9.127 GFLOP) in %.1f s: "
                      "oracle/plan_ref.py over oracle/conv_fast.cpp (im2col + the oracle's own AVX2-FMA GEMM where lele calls faer), 1 thread"
                      % (runs, ref.calls, runs * dt)}


def native_runner_leg(plan, blob, images, want, runs):
    """the SAME plan (folded: channel views, windows, conv2d_res; lane-scheduled when the DAG was taken) through lele_run, the native host
    (lele_amd/host: C++ over the C ABI, no Python, no torch): its own process, its own context, one recorded hipGraph replayed `runs` times"""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "lele_amd", "lele_run")
    if not os.path.exists(exe):
        return {"ran": False, "why": "lele_amd/lele_run is not built"}
    try:
        with tempfile.TemporaryDirectory() as d:
            json.dump(plan, open(os.path.join(d, "plan.json"), "w"))
            open(os.path.join(d, "weights.bin"), "wb").write(blob)
            images.tofile(os.path.join(d, "images.bin"))
            cmd = [exe, os.path.join(d, "plan.json"), os.path.join(d, "weights.bin"), "--out", os.path.join(d, "out"), "--input",
                   "images=%s:f32:%s" % (os.path.join(d, "images.bin"), ",".join(map(str, images.shape))), "--runs", str(runs), "--graph"]
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
            if r.returncode != 0:
                return {"ran": False, "why": r.stderr[-400:]}
            rec = json.loads(r.stdout.strip().splitlines()[-1])
            got = [np.fromfile(os.path.join(d, "out%d.bin" % k), np.float32).reshape(shape) for k, shape in enumerate(rec["outputs"])]
            return {"ran": True, "plan_format": plan.get("format"), "lanes": plan.get("dag", {}).get("lanes", 1), "kernel_calls": rec["kernel_calls"],
                    "graph_ms": round(rec["graph_ms"], 3), "eager_ms": round(rec["eager_ms"], 3),
                    "equals_python_runner_bitwise": bool(len(got) == len(want) and all(np.array_equal(a, b) for a, b in zip(got, want)))}
    except Exception as e:  # noqa: BLE001
        return {"ran": False, "why": "failed: %s" % e}


def yolo_leg(args, ctx, rank, world, fence, dist, device, comm=None, comm_note=None, ranks_seen=None):
    """BASELINE configs[4], batch `--yolo-batch` per GPU, one hipGraph a forward.  Two networks:
      * the reference's OWN generated Yolo26n-seg call sequence (examples/yolo26n-seg/src/yolo26seg.rs: 118 convolutions, 9.127 GFLOP an
        image), lifted into a plan (tools/lift_generated.py -> _lifted/yolo26seg_plan.json: an untracked artefact that travels with the
        working tree, not with a clone), its batch-1 shape literals re-batched, Concat / Split along C folded into channel views --
        the HEADLINE of this leg whenever the artefact is present;
      * the Yolo26n-seg-SHAPED network of tools/yolo_graph.py (same family, 100 convolutions, 9.736 GFLOP an image; compiled from ONNX by
        lele_amd.compiler) beside it -- the headline only in a checkout without the lifted plan.
    Synthetic weights.  Protocol of examples/yolo26n-seg/src/benchmark.rs:29-56: 3 warm-up forwards, then 10 timed ones.
    After the timed forwards: section 8(e)'s one exchange -- post-processing per image on the device (image.rs:127-265,
    lele_hip_yolo_seg_postprocess), then an all-gather of the fixed-width detection rows and their counts (RCCL through the C ABI)."""
    from lele_amd import kernels as K
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, fold_channel_views, load_weights_bin
    from lele_amd.sharded import all_gather_detections, all_gather_detections_rccl
    from lele_amd.tensor import TensorView
    from yolo_graph import yolo_onnx
    nb, size = args.yolo_batch, 640

    def bars(a, b):
        den = 1e-4 * np.maximum(np.abs(a), float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))) + 1e-7
        return float((np.abs(a - b) / den).max()) if a.size else 0.0

    def timed(runner, feed, gflop_per_image):
        """capture, 3 warm-up replays, `--yolo-runs` timed ones between fences; MAX over ranks of the wall time"""
        ctx.sync()
        ctx.graph_begin()
        outs = runner.run(feed)
        graph = ctx.graph_end()
        for _ in range(3):
            graph.launch()
        fence()
        t0 = time.perf_counter()
        ctx.timer_start()
        for _ in range(args.yolo_runs):
            graph.launch()
        ev_ms = ctx.timer_stop() / args.yolo_runs
        fence()
        wall = max_over_ranks(time.perf_counter() - t0, dist, device)
        ms = 1e3 * wall / args.yolo_runs
        tf = gflop_per_image * nb / ev_ms           # GFLOP / ms = TFLOP/s
        return graph, outs, {"runs": args.yolo_runs, "warmup": 3, "ms_per_forward": round(ms, 3), "ms_per_forward_hip_events_rank0": round(ev_ms, 3),
                             "images_per_s": round(world * nb * args.yolo_runs / wall, 1), "gflop_per_image": gflop_per_image,
                             "tflops_f32_per_gpu": round(tf, 2),
                             "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / F32_PEAK_TFLOPS, 4),
                                          "note": "convolution + attention multiply-adds of the whole forward (%.3f GFLOP an image) over the graph's HIP-event time, "
                                                  "against the f32-input MFMA peak SURVEY.md 8(d) names for configs[4].  The split-bf16 kernels (3 x 3 stride 1, "
                                                  "some 1 x 1) run on the bf16 matrix cores: six bf16 MFMAs a product, i.e. a ceiling of 2500 / 6 = 417 TFLOP/s "
                                                  "of f32-equivalent work for those layers" % gflop_per_image}}

    def as_dag(runner, feed, want):
        """the plan with its independent branches on lanes (lele_amd/lanes.py): per-statement DEVICE times (every statement recorded 8 times
        into a hipGraph and replayed), list scheduling over 4 lanes where a fork buys at least 40 us (a fork / join pair costs a hipGraph
        ~8 us: profiles/r05_dag_bench.json), buffers re-assigned under happens-before.  The scheduler's model lets lanes overlap perfectly,
        which kernels that fill the chip do not: what a statement costs a LANE beyond its device time (ramp, tail: a forward is 2 ms +
        0.1 ms an image, profiles/r06_yolo_fixed_costs.json) is tried at 0 / 15 / 30 / 50 us, every candidate recorded and replayed, the
        fastest kept.  A candidate is used only when its outputs equal `want` bit for bit."""
        from lele_amd.lanes import schedule
        try:
            runner.stmt_times, runner.stmt_repeat = [], 8
            runner.run(feed)
            times = {o: ms for _i, _fn, o, ms in runner.stmt_times}
            runner.stmt_times, runner.stmt_repeat = None, 1

            def replay_ms(r):
                ctx.sync()
                ctx.graph_begin()
                r.run(feed)
                g = ctx.graph_end()
                for _ in range(2):
                    g.launch()
                ctx.sync()
                ctx.timer_start()
                for _ in range(5):
                    g.launch()
                ms = ctx.timer_stop() / 5
                g.close()
                return ms
            linear_ms = replay_ms(runner)
            best, tried = None, []
            for extra in (0.0, 0.015, 0.030, 0.050):
                dag = schedule(runner.plan, {k: v + extra for k, v in times.items()}, lanes=4, min_gain_ms=0.04)   # (3 lanes: 7.83 ms, 4: 7.76 on the reference graph, tools/dag_bench.py)
                if dag is None or dag["dag"]["lanes"] < 2:
                    tried.append({"per_statement_lane_cost_us": round(extra * 1e3), "used": False, "why": "no branch worth a fork"})
                    continue
                r2 = Runner(dag, runner.raw, ctx)
                got = [o.numpy() for o in r2.run(feed)]
                if not all(np.array_equal(a, b) for a, b in zip(want, got)):
                    tried.append({"per_statement_lane_cost_us": round(extra * 1e3), "used": False, "why": "the DAG plan's outputs differ from the sequential plan's"})
                    r2.close()
                    continue
                ms = replay_ms(r2)
                tried.append({"per_statement_lane_cost_us": round(extra * 1e3), "statements_by_lane": dag["dag"]["statements_by_lane"], "events": dag["dag"]["events"],
                              "graph_ms": round(ms, 3)})
                if best is None or ms < best[0]:
                    if best is not None:
                        best[1].close()
                    best = (ms, r2, dag, extra)
                else:
                    r2.close()
            if best is None or best[0] >= linear_ms:
                if best is not None:
                    best[1].close()
                return runner, {"used": False, "why": "no lane schedule beats the linear graph (%.3f ms)" % linear_ms, "candidates": tried}
            return best[1], dict(best[2]["dag"], used=True, min_gain_ms=0.04, per_statement_lane_cost_us=round(best[3] * 1e3), candidates=tried,
                                 linear_graph_ms_when_chosen=round(linear_ms, 3))
        except Exception as e:  # noqa: BLE001
            runner.stmt_times, runner.stmt_repeat = None, 1
            ctx.lane_set(0)
            return runner, {"used": False, "why": "failed: %s" % e}

    # ---- the look-alike (always available)
    data, info = yolo_onnx(nb, size)
    plan, blob = compile_model(data, "yolo26n_seg_shaped_n%d" % nb)
    weights = load_weights_bin(plan, blob)
    images = np.random.default_rng(1000 + rank).uniform(0, 1, (nb, 3, size, size)).astype(np.float32)   # uniform[0, 1)
    feed = {"images": TensorView(ctx.buf().upload(images))}
    r0 = Runner(plan, weights, ctx)
    r0.shapes = {}
    base = [o.numpy().copy() for o in r0.run(feed)]
    look = {"model": "Yolo26n-seg-SHAPED (tools/yolo_graph.py), synthetic weights", "batch_per_gpu": nb, "input": [nb, 3, size, size], **info,
            "plan_calls": r0.calls}
    layers = yolo_conv_layers(plan, r0.shapes) if rank == 0 else None
    runner = r0
    try:
        folded = fold_channel_views(plan, r0.shapes)
        r1 = Runner(folded, weights, ctx)
        got = [o.numpy() for o in r1.run(feed)]
        if not all(np.array_equal(a, b) for a, b in zip(base, got)):
            raise RuntimeError("the folded plan's outputs differ from the plan's")
        for b_ in r0.ws.values():
            b_.close()
        runner = r1
        look.update({"channel_views": folded["folded"], "plan_calls": r1.calls, "folded_equals_unfolded_bitwise": True})
    except Exception as e:  # noqa: BLE001  -- the leg is still measured, on the unfolded plan, and says so
        look["channel_views"] = "channel views NOT used: %s" % e
    if not args.no_dag:
        graph, _o, t_lin = timed(runner, feed, info["gflop_per_image"])
        graph.close()
        runner, look["dag"] = as_dag(runner, feed, base)
        look["dag"]["linear_graph_ms"] = t_lin["ms_per_forward_hip_events_rank0"]
    graph, outs, t = timed(runner, feed, info["gflop_per_image"])
    look.update(t)
    look.update({"graph_equals_eager_bitwise": bool(all(np.array_equal(a, o.numpy()) for a, o in zip(base, outs))),
                 "finite": bool(all(np.isfinite(a).all() for a in base))})
    if rank == 0:
        # two images of the batch against the batch-1 plan of the same network (same seed -> same weights): the prototype map value for
        # value, the detections' scores in order.  (Against the CPU ORACLE's forward: tests/test_graph_oracle.py.)
        d1, _ = yolo_onnx(1, size)
        p1, b1 = compile_model(d1, "yolo26n_seg_shaped_n1")
        one = Runner(p1, load_weights_bin(p1, b1), ctx)
        x1 = ctx.buf()
        worst = 0.0
        for i in (0, nb - 1):
            o1 = [o.numpy() for o in one.run({"images": TensorView(x1.upload(images[i:i + 1]))})]
            for a, b in zip(o1, base):
                worst = max(worst, bars(a[..., 4], b[i:i + 1][..., 4]) if a.ndim == 3 else bars(a, b[i:i + 1]))
        look.update({"images_checked_against_the_batch_1_plan": 2, "max_error_in_units_of_1e-4": round(worst, 4), "per_image_check_ok": bool(worst <= 1.0)})
    graph.close()
    if rank == 0 and world == 1 and not args.no_native:
        look["native_runner"] = native_runner_leg(runner.plan, blob, images, base, args.yolo_runs)
    head, head_outs, head_name = look, outs, "look-alike"

    # ---- the reference's own generated graph, where the lifted plan is present (every rank runs it on its own images)
    lifted = os.path.join(ROOT, "_lifted", "yolo26seg_plan.json")
    ref = None
    if os.path.exists(lifted):
        try:
            import yolo_lifted_batch as Y
            one, big, lfeed, limages, lname, louts, ref = Y.build(ctx, lifted, nb, seed=1000 + rank)
            worst = 0.0
            if rank == 0:
                lx1 = ctx.buf()
                for i in (0, nb - 1):
                    o1 = [o.numpy() for o in one.run({lname: TensorView(lx1.upload(limages[i:i + 1]))})]
                    for a, b in zip(o1, louts):
                        worst = max(worst, bars(a[..., 4], b[i:i + 1][..., 4]) if a.ndim == 3 else bars(a, b[i:i + 1]))
            if not args.no_dag:
                lgraph, _o, t_lin = timed(big, lfeed, 9.127)
                lgraph.close()
                big, ref["dag"] = as_dag(big, lfeed, louts)
                ref["dag"]["linear_graph_ms"] = t_lin["ms_per_forward_hip_events_rank0"]
            lgraph, lres, t = timed(big, lfeed, 9.127)
            ref.update(t)
            ref.update({"graph_equals_eager_bitwise": bool(all(np.array_equal(a, o.numpy()) for a, o in zip(louts, lres))),
                        "max_error_in_units_of_1e-4_vs_the_batch_1_plan": round(worst, 4), "per_image_check_ok": bool(worst <= 1.0)})
            lgraph.close()
            head, head_outs, head_name = ref, lres, "reference"
        except Exception as e:  # noqa: BLE001
            ref = {"failed": str(e)}
    rec = dict(head)
    rec["graph"] = ("the reference's own generated Yolo26n-seg call sequence (examples/yolo26n-seg/src/yolo26seg.rs, lifted and re-batched)"
                    if head_name == "reference" else
                    "Yolo26n-seg-SHAPED look-alike (no _lifted/yolo26seg_plan.json in this checkout: tools/lift_generated.py lift makes it where the reference is mounted)")
    if head_name == "reference":
        rec["lookalike"] = {k: look[k] for k in ("model", "convolutions", "gflop_per_image", "plan_calls", "dag", "ms_per_forward", "ms_per_forward_hip_events_rank0",
                                                 "images_per_s", "tflops_f32_per_gpu", "roofline", "max_error_in_units_of_1e-4", "per_image_check_ok",
                                                 "graph_equals_eager_bitwise", "native_runner") if k in look}
    elif ref is not None:
        rec["reference_graph"] = ref
    # parity of what runs here (tests/): convolutions 1e-4 against the oracle; whole graphs at batch 64 against the oracle's forward
    # (tests/test_graph_oracle.py: prototype map, pre-top-k tensors, detections row by row)
    rec["parity"] = {"conv_bar": 1e-4, "graph_bar": 1e-4,
                     "conv_epilogue_silu": "replica" if os.environ.get("LELE_HIP_CONV_SILU_EXACT", "0") not in ("", "0")
                     else "v_exp_f32 / v_rcp_f32 + one Newton step, <= 1e-5 relative + 1e-7 of the reference's"}

    # ---- section 8(e), "C5": collect the batch's outputs.  Per image on the device: score / box filter and mask (image.rs:127-265);
    # what crosses a link is [300, 38] rows + a count per image.  One step = post-processing + both all-gathers + the read-back.
    total = world * nb
    gbufs = [ctx.buf() for _ in range(4)]
    pbufs = [ctx.buf() for _ in range(3)]

    def collect():
        dets, counts, _mask = K.yolo_seg_postprocess(head_outs[0], head_outs[1], size, size, 0.25, 80, out_dets=pbufs[0], out_count=pbufs[1],
                                                     out_mask=pbufs[2], ctx=ctx)
        if comm is not None:
            return all_gather_detections_rccl(dets, counts, total, comm, ctx, gbufs)
        if dist is not None:
            return all_gather_detections(dets.numpy(), counts.numpy(), total, dist, device)
        return all_gather_detections(dets.numpy(), counts.numpy(), total)
    everything = collect()
    fence()
    t0 = time.perf_counter()
    for _ in range(3):
        everything = collect()
    fence()
    gather_ms = 1e3 * max_over_ranks(time.perf_counter() - t0, dist, device) / 3
    mine = everything[rank * nb:(rank + 1) * nb]
    local = all_gather_detections(*[a.numpy() for a in K.yolo_seg_postprocess(head_outs[0], head_outs[1], size, size, 0.25, 80, ctx=ctx)[:2]], nb)
    rec["gather"] = {"what": "per-image detections after lele_hip_yolo_seg_postprocess (threshold 0.25): f32 [images, 300, 38] rows + i32 counts",
                     "collective": ("rccl all-gather via lele_hip_comm_allgather + lele_hip_comm_allgather_i32" if comm is not None
                                    else (comm_note or "none (single process)")),
                     "images": len(everything), "images_expected": total, "bytes_per_rank": nb * (300 * 38 * 4 + 4),
                     "postprocess_gather_readback_ms": round(gather_ms, 3),
                     "kept_detections_per_rank": [int(sum(d.shape[0] for d in everything[r * nb:(r + 1) * nb])) for r in range(world)],
                     "own_block_equals_local_postprocess": bool(len(mine) == len(local) and all(np.array_equal(a, b) for a, b in zip(mine, local)))}
    if ranks_seen is not None:
        rec["gather"]["rccl_ranks_seen"] = ranks_seen
    if rank == 0:
        rec["_layers"] = layers
    return rec


def run_rank(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d: refusing to report n_gpus that is not the number of ranks running"
                         % (args.gpus, world))
    if rank != 0:   # only rank 0 reports; whatever the libraries of the other ranks print to stdout (RCCL's banner) must not reach the caller
        silence_stdout()
    if args.dry_run:  # launcher logic only (CPU tests): rendezvous over gloo, the fences and the MAX all-reduce, no GPU work
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        dist.barrier()
        wall = max_over_ranks(0.001 * (rank + 1), dist, "cpu")
        # what every rank WOULD open and process: its device (LOCAL_RANK, one per rank), its shard of both sharded legs
        mine = {"rank": rank, "device": local_rank, "utterances": list(shard_range(args.per_gpu * world, rank, world)),
                "images": [rank * args.yolo_batch, (rank + 1) * args.yolo_batch]}
        seen = [None] * world
        dist.all_gather_object(seen, mine)
        dist.barrier()
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "value": wall,
                              "ranks_seen": len(seen), "devices": [m["device"] for m in seen],
                              "one_device_per_rank": len({m["device"] for m in seen}) == world,
                              "utterance_shards": [m["utterances"] for m in seen], "image_shards": [m["images"] for m in seen]}), flush=True)
        dist.destroy_process_group()
        return
    dist, device = None, "cpu"
    if world > 1 or os.environ.get("LELE_BENCH_FORCE_DIST") == "1":  # FORCE_DIST: exercise RCCL init + collectives at N=1
        import torch
        import torch.distributed as dist  # backend "nccl" is RCCL on ROCm
        # LELE_BENCH_BACKEND=gloo + LELE_BENCH_SHARE_GPU=1 exist only to exercise this N>1 path on a one-GPU box
        backend = os.environ.get("LELE_BENCH_BACKEND", "nccl")
        if os.environ.get("LELE_BENCH_SHARE_GPU") == "1":
            local_rank = local_rank % max(1, torch.cuda.device_count())
            # several PROCESSES on one device: the one-launch feed-forward kernel waits for workgroups of its own launch, which another
            # process's kernels may keep off the CUs -- the two-launch form where ranks share a GPU
            os.environ.setdefault("LELE_HIP_FFN_ONE_LAUNCH", "0")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            device = torch.device("cuda", local_rank)
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    import lele_amd
    ctx = lele_amd._lib.Ctx(local_rank)

    def fence():
        ctx.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    # The recogniser legs run first: they are seconds of sustained device work, so the front-end's W warm-up + K timed steps --
    # half a millisecond each -- then run at settled clocks even when the caller asks for few steps (a cold 20-step run reads
    # 0.55-0.57 ms per step where the steady state is 0.48).
    sv = None
    yo = None
    c0 = None
    if rank == 0 and world == 1 and not args.no_model:
        try:
            c0 = configs0_leg(ctx)
        except Exception as e:  # noqa: BLE001
            c0 = {"failed": str(e)}
    comm, comm_note, ranks_seen = open_comm(args, ctx, rank, world, dist, device)
    if not args.no_model:
        sv = sensevoice_leg(args, ctx, rank, world, fence, dist, device, comm, comm_note, ranks_seen)
    if not args.no_yolo:
        yo = yolo_leg(args, ctx, rank, world, fence, dist, device, comm, comm_note, ranks_seen)
    if comm is not None:
        comm.close()
    fe = frontend_leg(args, ctx, rank, world, fence, dist, device)

    if rank == 0:
        n, wall = fe["n"], fe["wall"]
        total_bytes = world * args.batch * fe["bytes_per_utt"] * args.steps
        value = total_bytes / wall / 1e9
        audio_s = world * args.batch * SECONDS * args.steps
        main_ms = fe["main_ms"]
        achieved = args.batch * fe["bytes_per_utt"] / (main_ms * 1e-3) / 1e9 if main_ms > 0 else 0.0
        prof = {}
        pf = os.path.join(ROOT, "profiles", "frontend_roofline.json")
        if os.path.exists(pf):  # PMC / microbenchmark constants measured on the box (profiles/README.md): traffic, VALU work and peak
            try:
                prof = json.load(open(pf))
            except Exception:
                prof = {}
        # the profile's constants are per launch of prof["batch"] utterances; both scale linearly with the batch
        pscale = args.batch / float(prof["batch"]) if prof.get("batch") else 0.0
        # the governing roofline of STFT+mel is HBM -- `frac` = algorithmic bytes / kernel time / 8 TB/s.  Beside it:
        # `compute_frac` = the SURVEY's algorithmic flop count (sparse mel: 0.075 GFLOP per 30 s utterance) / kernel time / the f32
        # vector peak, and `valu_issue_frac` = issued VALU lane-operations (PMC SQ_INSTS_VALU x the loop's ISA mix) / the measured
        # issue ceiling: an issue-EFFICIENCY figure (overhead instructions count as work), not a roofline fraction.
        flop_per_utt = 0.075e9 * (n / float(SAMPLE_RATE * 30))
        roof = {"kernel": "fe_main_kernel", "kernel_ms": round(main_ms, 5), "launches": fe["runs"],
                "algorithmic_bytes_per_launch": args.batch * fe["bytes_per_utt"],
                "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": int(prof["hbm_bytes_per_launch"] * pscale) if pscale and prof.get("hbm_bytes_per_launch") else None,
                "compute_frac": round(args.batch * flop_per_utt / (main_ms * 1e-3) / (F32_PEAK_TFLOPS * 1e12), 4) if main_ms > 0 else None,
                "compute_frac_source": "SURVEY.md 8(d): 0.075 GFLOP per 30 s utterance (FFT 23 040 flop + sparse mel) / kernel time / 157.3 TFLOP/s"}
        # What actually governs this kernel is not HBM: at 8 TB/s its bytes are ~0.8 ms of a 3.4 ms launch; its issued VALU
        # lane-operations at the measured issue ceiling are ~2.4 ms.  `bound` / `frac` stay the figures SURVEY.md 8(d) defines (HBM);
        # `governed_by` names the pipe the kernel is limited by and `governing_frac` its share of that pipe's measured ceiling.
        lane_ops = int(prof["valu_lane_ops_per_launch"] * pscale) if pscale and prof.get("valu_lane_ops_per_launch") else None
        valu_peak = prof.get("valu_peak_lane_ops_per_s")
        if lane_ops and valu_peak and main_ms > 0:
            a = lane_ops / (main_ms * 1e-3)
            roof.update({"governed_by": "valu-issue", "governing_frac": round(a / valu_peak, 4),
                         "governing_floor_ms": round(lane_ops / valu_peak * 1e3, 4), "hbm_floor_ms": round(args.batch * fe["bytes_per_utt"] / (HBM_PEAK_GBS * 1e9) * 1e3, 4),
                         "valu_issue_frac": round(a / valu_peak, 4), "valu_lane_ops_per_launch": lane_ops,
                         "valu_issue_achieved_tlane_ops": round(a / 1e12, 3), "valu_issue_peak_tlane_ops": round(valu_peak / 1e12, 3),
                         "valu_issue_source": "issued instructions (rocprofv3 SQ_INSTS_VALU x ISA mix, tools/summarize_profile.py) over the "
                                              "ceiling measured by tools/valu_rate.hip (%s)" % prof.get("valu_peak_source"),
                         "valu_issue_frac_datasheet": round(a / 1e12 / VALU_PEAK_TLANE, 4)})
        yolo_workload = "not run" if yo is None else ("the reference's own generated graph, lifted and re-batched; the look-alike beside it"
                                                     if "lookalike" in yo else "the Yolo26n-seg-shaped look-alike: no lifted plan in this checkout")
        line = {
            "metric": "STFT+mel GB/s (SenseVoice front-end: PCM -> log-mel -> LFR, algorithmic bytes); SenseVoiceSmall-shaped RTF in `sensevoice`",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: STFT+mel+LFR, 30 s synthetic 16 kHz mono, batch %d utterances per GPU per step; "
                                   "configs[0] (Silero-shaped streaming chain on zh.wav) in `configs0`, configs[2]/[3] SenseVoice-shaped recogniser in `sensevoice`, configs[4] Yolo26n-seg at batch 64 per GPU in `yolo` (%s)" % (args.batch, yolo_workload),
                       "samples_per_utterance": n, "frames": fe["nf"], "lfr_rows": fe["t_lfr"], "batch_per_gpu": args.batch,
                       "bytes_per_utterance": fe["bytes_per_utt"], "parallelism": "utterance-sharded x%d" % world},
            "rtf_frontend": round(wall / audio_s, 9),
            "roofline": roof,
        }
        enc = feats = dev_logits = None
        if sv is not None:
            feats, enc, dev_logits = sv.pop("_c3_feats", None), sv.pop("_enc", None), sv.pop("_c3_logits", None)
            line["sensevoice"] = sv
            for k in ("rtf_model", "rtf_e2e", "rtf_c4", "audio_s_per_s", "rtf_model_exact", "rtf_c4_exact"):
                if k in sv:
                    line[k] = sv[k]
            q = sv.get("qlinear")
            if q and q.get("op_ms"):
                # flat copies inside `roofline`: the model path's dominant kernel, priced against HBM (its bound at K = 512: 57 MB of
                # f32 traffic = 7 us at 8 TB/s against 2.9 us of i8 MFMA) and, beside it, against the i8 matrix-core peak
                roof.update({"model_kernel": "fused_quantized_linear " + q["shape"], "model_bound": "hbm", "model_op_ms": q["op_ms"],
                             "model_achieved": q["hbm_gbs"], "model_peak": HBM_PEAK_GBS, "model_unit": "GB/s",
                             "model_frac": round(q["hbm_gbs"] / HBM_PEAK_GBS, 4), "model_tops": q["tops"],
                             "model_mfma_frac": round(q["tops"] / I8_PEAK_TOPS, 4)})
            fb = sv.get("ffn_block")
            if fb and fb.get("op_ms"):   # the fused kernels of round 6 (two launches for the whole feed-forward block of a layer)
                roof.update({"model_block": fb["what"], "model_block_op_ms": fb["op_ms"], "model_block_frac": round(fb["hbm_gbs"] / HBM_PEAK_GBS, 4),
                             "model_block_tops": fb["tops"], "model_block_mfma_frac": round(fb["tops"] / I8_PEAK_TOPS, 4)})
        layers = None
        if yo is not None:
            layers = yo.pop("_layers", None)
            line["yolo"] = yo
            for k in ("ms_per_forward", "images_per_s"):
                line["yolo_" + k] = yo[k]
        if c0 is not None:
            dev_probs = c0.pop("_probs", None)
            line["configs0"] = c0
            if dev_probs is not None and not args.no_cpu_baseline:
                line["configs0"]["cpu_baseline"] = cpu_baseline_configs0(dev_probs)
        if world == 1 and not args.no_cpu_baseline:  # reported baseline: rank 0 at N=1 only
            lifted = os.path.join(ROOT, "_lifted", "yolo26seg_plan.json")
            if yo is not None and "lookalike" in yo and os.path.exists(lifted):
                line["yolo"]["cpu_baseline"] = cpu_baseline_yolo_graph(lifted)
                if layers:
                    line["yolo"]["lookalike"]["cpu_baseline"] = cpu_baseline_yolo(layers)
            elif layers:
                line["yolo"]["cpu_baseline"] = cpu_baseline_yolo(layers)
            cb = cpu_baseline_frontend(n)
            if enc is not None and feats is not None:
                from sensevoice_graph import encoder_arrays
                cb.update(cpu_baseline_model(encoder_arrays(enc), feats, dev_logits=dev_logits))
                if "oracle_agreement" in cb:
                    line["sensevoice"]["oracle_agreement"] = cb.pop("oracle_agreement")
            line["cpu_baseline"] = cb
            line["cpu_baseline_all_cores"] = cpu_baseline_frontend_all_cores(n)
            if enc is not None and feats is not None:
                line["cpu_baseline_all_cores"].update(cpu_baseline_model_all_cores(encoder_arrays(enc), feats))
        emit_last_line(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def silence_stdout():
    """everything the C libraries still hold in their stdio buffers (RCCL prints a version banner to stdout when a communicator
    is created; it is flushed at exit) goes out NOW, and nothing written to stdout afterwards reaches the caller"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    devnull = os.open(os.devnull, os.O_WRONLY)
    os.dup2(devnull, 1)
    os.close(devnull)


def emit_last_line(text):
    """the bench contract: rank 0 prints ONE JSON line -- make it the LAST thing on stdout"""
    import ctypes
    sys.stdout.flush()
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    os.write(1, (text + "\n").encode())
    silence_stdout()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn(args, argv):
    """--gpus N > 1 without a launcher: start the N ranks ourselves (one process per GPU) and pass rank 0's line through"""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    while any(p.poll() is None for p in procs):  # a rank that dies takes the job with it (the others would wait in a collective)
        for p in procs:
            if p.poll() not in (None, 0):
                rc = p.returncode
        if rc:
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        rc = p.wait() or rc
    return rc


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=2048, help="front-end leg: 30 s utterances per GPU per step (6.2 GB of PCM + features resident in "
                    "HBM; a step is then ~3.7 ms, so that few-step runs are not dominated by the device's ~25 ms clock ramp)")
    ap.add_argument("--per-gpu", type=int, default=32, help="recogniser leg: 10 s utterances per GPU per step (configs[3]: 256 / 8)")
    ap.add_argument("--sv-steps", type=int, default=10, help="recogniser leg: timed steady-state steps (lele's harness uses 10)")
    ap.add_argument("--layers", type=int, default=70)
    ap.add_argument("--no-model", action="store_true", help="skip the SenseVoice-shaped recogniser legs")
    ap.add_argument("--no-yolo", action="store_true", help="skip the configs[4] leg")
    ap.add_argument("--no-dag", action="store_true", help="configs[4]: record the plans as linear graphs (default: independent branches on lanes, lele_amd/lanes.py)")
    ap.add_argument("--gather-logits", action="store_true", help="configs[3]: also exchange the FULL logits (SURVEY.md 8(e): 17.1 MB an utterance), timed once")
    ap.add_argument("--no-native", action="store_true", help="configs[4]: skip the replay of the look-alike's plan through the native host (lele_amd/lele_run)")
    ap.add_argument("--yolo-batch", type=int, default=64, help="configs[4]: 640 x 640 images per GPU per forward")
    ap.add_argument("--yolo-runs", type=int, default=10, help="configs[4]: timed forwards after 3 warm-up ones (examples/yolo26n-seg/src/benchmark.rs:29-56)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-fallback", action="store_true", help="N > 1: if the C ABI's RCCL communicator cannot be set up on every rank, gather the "
                    "ids with torch.distributed instead of failing (the line then names that transport)")
    ap.add_argument("--dry-run", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(spawn(args, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
