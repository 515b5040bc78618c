#!/usr/bin/env python3
"""bench.py -- STFT+mel front-end throughput on MI355X (BASELINE.json configs[1], batched).

A "step" = one pass of the hot path (SenseVoiceFrontend: PCM -> log-mel -> LFR) over one batch of
`--batch` synthetic 30 s / 16 kHz utterances already resident in HBM.  One process per GPU; each rank owns its
own batch (utterances are independent: weak scaling, no data-path collective).

Prints ONE JSON line (rank 0):
  value     = algorithmic GB/s of the whole job = ranks * batch * (4*S + 4*T*560) bytes * steps / wall time
  roofline  = the dominant kernel (fe_main_kernel) against the 8 TB/s HBM peak, timed with HIP events on the
              stream it is launched on (lele_hip_frontend_set_profiling)
  cpu_baseline = the CPU oracle (C++ restatement of lele's x86 AVX2 path, 1 thread = lele's execution model)
              timed on this host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SAMPLE_RATE = 16000
SECONDS = 30
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def synth_batch(batch, n, seed0):
    """0.3 sin(2pi 220 t) + 0.2 sin(2pi 1000 t) + 0.05 U(-1,1), seed per utterance"""
    t = np.arange(n) / float(SAMPLE_RATE)
    tone = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1000 * t)).astype(np.float32)
    out = np.empty((batch, n), np.float32)
    for i in range(batch):
        rng = np.random.default_rng(seed0 + i)
        out[i] = tone + (0.05 * rng.uniform(-1, 1, n)).astype(np.float32)
    return out


def cpu_baseline_all_cores(n, budget_s=6.0):
    """lele's only route to multi-core is one independent instance per core (everything in it is
    Par::Seq with thread-local scratch), utterances sharded round-robin.  One forked worker PROCESS per hardware thread,
    each looping over the oracle's front-end for `budget_s` seconds (separate address spaces: threads of one process
    serialise on the allocator's mmap traffic and scale only ~6x on 256 cores)."""
    from oracle import pyoracle as O
    O.lib()
    hw = os.cpu_count() or 1
    cores = hw
    try:  # the container may be granted fewer CPUs than the host has (cgroup v2 cpu.max = "quota period")
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            cores = max(1, min(hw, -(-int(q) // int(per))))
    except Exception:
        pass
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
            "sample": "%d x 30 s utterances over %d independent single-threaded oracle processes (CPU quota of this "
                      "container: %d of the host's %d hardware threads), %.1f s wall" % (total, cores, cores, hw, el),
            "rtf": round(el / max(1, total) / (n / SAMPLE_RATE), 7)}


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


def cpu_baseline(n, budget_s=12.0):
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
            "sample": "%d x 30 s utterances, %.1f s wall, oracle/liboracle.so (C++ restatement of lele's x86 AVX2 "
                      "path), single thread; host has %d cores" % (done, el, os.cpu_count() or 0),
            "rtf": round(el / (done * n / SAMPLE_RATE), 6)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("LELE_BENCH_FORCE_DIST") == "1":  # FORCE_DIST: exercise RCCL init + collectives at N=1
        import torch
        import torch.distributed as dist  # backend "nccl" is RCCL on ROCm
        # LELE_BENCH_BACKEND=gloo + LELE_BENCH_SHARE_GPU=1 exist only to exercise this N>1 path on a one-GPU box
        backend = os.environ.get("LELE_BENCH_BACKEND", "nccl")
        if os.environ.get("LELE_BENCH_SHARE_GPU") == "1":
            local_rank = local_rank % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import lele_amd
    from lele_amd.features import SenseVoiceFrontend

    ctx = lele_amd._lib.Ctx(local_rank)
    fe = SenseVoiceFrontend(ctx=ctx)
    n = SAMPLE_RATE * SECONDS
    t_lfr, cols, nf = fe.out_rows(n)
    bytes_per_utt = 4 * n + 4 * t_lfr * cols  # PCM read once + LFR written once
    # weak scaling: the global batch is world * batch utterances; this rank synthesises and keeps its own shard
    lo, hi = shard_range(world * args.batch, rank, world)
    pcm = ctx.buf().upload(synth_batch(hi - lo, n, rank_seed_base(rank, world * args.batch, world)))  # resident in HBM
    out = ctx.buf()

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        fe.compute_batch(pcm, out)
    barrier()
    fe.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fe.compute_batch(pcm, out)
    ctx.sync()
    barrier()
    wall = time.perf_counter() - t0
    sum_ms, main_ms, runs = fe.profile_read()
    fe.set_profiling(False)
    wall = max_over_ranks(wall, dist, "cuda" if dist is None or dist.get_backend() == "nccl" else "cpu")

    if rank == 0:
        total_bytes = world * args.batch * bytes_per_utt * args.steps
        value = total_bytes / wall / 1e9
        audio_s = world * args.batch * SECONDS * args.steps
        achieved = args.batch * bytes_per_utt / (main_ms * 1e-3) / 1e9 if main_ms > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "frontend_hbm_traffic.json")
        if os.path.exists(tf):  # PMC-measured HBM bytes per fe_main_kernel launch at this batch (see profiles/README.md)
            try:
                rec = json.load(open(tf))
                if rec.get("batch") == args.batch:
                    traffic = rec.get("bytes_per_launch")
            except Exception:
                traffic = None
        valu_busy = None
        pf = os.path.join(ROOT, "profiles", "r01_frontend_pmc_sq.json")
        if os.path.exists(pf):  # SQ counters of the same kernel (profiles/README.md): the kernel is VALU-bound, not HBM-bound
            try:
                c = json.load(open(pf))
                valu_busy = round(c["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * c["GRBM_GUI_ACTIVE"] / 8), 3)
            except Exception:
                valu_busy = None
        line = {
            "metric": "STFT+mel GB/s (SenseVoice front-end: PCM -> log-mel -> LFR, algorithmic bytes)",
            "value": round(value, 3), "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(wall / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: STFT+mel+LFR, 30 s synthetic 16 kHz mono, batch %d "
                                   "utterances per GPU per step" % args.batch,
                       "samples_per_utterance": n, "frames": nf, "lfr_rows": t_lfr, "batch_per_gpu": args.batch,
                       "bytes_per_utterance": bytes_per_utt, "parallelism": "utterance-sharded x%d" % world},
            "rtf": round(wall / audio_s, 9),
            "roofline": {"bound": "hbm", "kernel": "fe_main_kernel", "achieved": round(achieved, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel_ms": round(main_ms, 5),
                         "aux_kernel": "none (frame sums are fused into fe_main_kernel; LELE_HIP_FE_FUSED=0 restores "
                                       "the separate fe_frame_sum_kernel)" if sum_ms < 0.02 else "fe_frame_sum_kernel",
                         "aux_kernel_ms": round(sum_ms, 5), "launches": runs,
                         "algorithmic_bytes_per_launch": args.batch * bytes_per_utt,
                         "valu_busy_frac_pmc": valu_busy,
                         "note": "bit-exact radix-2 FFT replica: ~19 VALU lane-ops per algorithmic byte against a ridge of ~4.9 "
                                 "-> VALU-bound by construction (DESIGN.md 3.1); traffic is an upper bound (profiles/README.md)"},
        }
        if world == 1 and not args.no_cpu_baseline:  # reported baseline: rank 0 at N=1 only
            line["cpu_baseline"] = cpu_baseline(n)
            line["cpu_baseline_all_cores"] = cpu_baseline_all_cores(n)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
