#!/usr/bin/env python3
"""SURVEY.md section 8(e) end to end: the SenseVoice-shaped recogniser, one process per GPU.

    python tools/sensevoice_sharded.py                                   # one GPU (a 1-rank RCCL group with --dist)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/sensevoice_sharded.py --dist

Rank r owns utterances [lo, hi) of a global batch of `--per-gpu` x N ten-second utterances (weak scaling, C4 at N = 8),
weights replicated (same seed on every rank).  A step is: batched front-end (PCM -> LFR log-mel) -> CMVN -> the compiled
plan of the encoder replayed as one hipGraph -> greedy arg-max and blank / special-token filter on the device -> one
all-gather of the token ids (`lele_amd.sharded.all_gather_ids`; 22 KB per GPU) so that every rank holds the transcripts of
the whole batch.  No other traffic crosses a link.  Timed region: barrier + device sync on both sides, MAX over ranks.
The topology is assumed and the weights synthetic (SURVEY.md 8a note); what is measured is the path, not accuracy."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--per-gpu", type=int, default=32)
    ap.add_argument("--seconds", type=int, default=10)
    ap.add_argument("--layers", type=int, default=70)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--dist", action="store_true", help="initialise torch.distributed (nccl = RCCL) even for one rank")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    rank, world, local = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1)), int(os.environ.get("LOCAL_RANK", 0))
    dist, device = None, "cpu"
    if args.dist or world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        device = torch.device("cuda", local)
        torch.cuda.set_device(device)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
    import bench
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd.compiler import compile_model
    from lele_amd.features import Cmvn, SenseVoiceFrontend
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.sharded import all_gather_ids, shard_range
    from sensevoice_graph import VOCAB, Encoder, encoder_onnx

    total = args.per_gpu * world
    lo, hi = shard_range(total, rank, world)
    n = 16000 * args.seconds
    ctx = lele_amd._lib.Ctx(local)
    fe, cmvn = SenseVoiceFrontend(ctx=ctx), Cmvn(ctx=ctx)
    plan, blob = compile_model(encoder_onnx(Encoder(ctx, args.layers), hi - lo), "sensevoice_shaped")
    runner = Runner(plan, load_weights_bin(plan, blob), ctx)
    skip = np.zeros(VOCAB, np.uint8)          # blank + a block of <|...|> specials, as tokenizer.rs:38-48 marks them
    skip[0] = 1
    skip[VOCAB - 200:] = 1
    skip = lele_amd._lib.Weight(skip)
    pcm = ctx.buf().upload(bench.synth_batch(hi - lo, n, lo))     # utterance i is synthesised from seed i on whichever rank owns it
    fbuf, cbuf, abuf, ibuf, nbuf = (ctx.buf() for _ in range(5))

    def features():
        return cmvn.compute(fe.compute_batch(pcm, fbuf), out=cbuf)

    feats = features()
    runner.run({"feats": feats})
    ctx.sync()
    ctx.graph_begin()
    logits = runner.run({"feats": feats})[0]
    graph = ctx.graph_end()

    def step():
        features()                                                  # same buffers every step: the graph reads cbuf
        graph.launch()
        ids, counts = K.token_filter(K.argmax_last(logits, out=abuf, ctx=ctx), skip, out_ids=ibuf, out_counts=nbuf, ctx=ctx)
        return all_gather_ids(ids.numpy(), counts.numpy(), total, dist, device)   # .numpy() waits for the stream

    def fence():
        ctx.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        everything = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        everything = step()
    fence()
    wall = bench.max_over_ranks(time.perf_counter() - t0, dist, device)
    # every rank must hold the same transcripts, and its own block of them must be what it decoded itself
    mine = all_gather_ids(*[a.numpy() for a in K.token_filter(K.argmax_last(logits, ctx=ctx), skip, ctx=ctx)], hi - lo)
    own_ok = all(np.array_equal(a, b) for a, b in zip(everything[lo:hi], mine))
    digest = int(sum(int(np.int64(a).sum()) * (i + 1) for i, a in enumerate(everything)) % (1 << 61))
    if dist is not None:
        import torch
        d = torch.tensor([digest, -digest, int(own_ok)], dtype=torch.int64, device=device)
        dist.all_reduce(d, op=dist.ReduceOp.MIN)
        agree = bool(d[0].item() == digest and -d[1].item() == digest and d[2].item() == 1)
    else:
        agree = own_ok
    if rank == 0:
        audio = total * args.seconds
        rec = {"metric": "sensevoice_shaped_rtf_frontend_model_decode_gather", "value": round(wall / args.steps / audio, 8), "unit": "s/s",
               "higher_is_better": False, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * wall / args.steps, 3), "scaling": "weak", "utterances": total, "utterances_per_gpu": args.per_gpu,
               "seconds_per_utterance": args.seconds, "layers": args.layers, "collective": "rccl all-gather of token ids" if dist else "none (single process)",
               "gathered_bytes_per_gpu": int((hi - lo) * (1 + logits.shape[1]) * 4), "tokens_kept_first_utterance": int(len(everything[0])),
               "ranks_agree": agree, "data": "synthetic PCM, synthetic weights, assumed topology"}
        print(json.dumps(rec), flush=True)
        if args.out:
            os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
            json.dump(rec, open(args.out, "w"), indent=1)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not agree:
        raise SystemExit("ranks disagree on the gathered token ids")


if __name__ == "__main__":
    main()
