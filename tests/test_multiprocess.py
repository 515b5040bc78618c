"""Multi-GPU path: utterances shard across ranks with no data-path collective; the only
communication is the barrier pair around the timed region and the MAX all-reduce of the wall time.  bench.py's N>1 leg
runs that logic over RCCL; here the same functions run over gloo with world_size 2 on CPU, each rank processing its
shard with the CPU oracle in place of the device front-end."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ["LELE_ROOT"])
import torch
import torch.distributed as dist
import bench
from oracle import pyoracle as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total, n = 6, 16000
lo, hi = bench.shard_range(total, rank, world)
pcm = bench.synth_batch(hi - lo, n, bench.rank_seed_base(rank, total, world))
dist.barrier()
t0 = time.perf_counter()
feats = np.stack([O.frontend_compute(p) for p in pcm])
dist.barrier()
wall = bench.max_over_ranks(time.perf_counter() - t0, dist, device="cpu")
# collect the shard checksums only to let the test verify coverage (not part of the data path)
sums = [None] * world
dist.all_gather_object(sums, (lo, hi, [float(np.float64(f).sum()) for f in feats]))
if rank == 0:
    print(json.dumps({"wall": wall, "shards": sums}))
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_shard_range_partitions_the_batch():
    import bench
    for total in (1, 7, 8, 256, 257):
        for world in (1, 2, 3, 8):
            spans = [bench.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_run_matches_single_process(tmp_path):
    import bench
    from oracle import pyoracle as O
    port = _free_port()
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), LELE_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    rec = json.loads(outs[0][0].strip().splitlines()[-1])
    assert rec["wall"] > 0
    # every utterance processed exactly once, with the same result a single process gets for that utterance
    got = {}
    for lo, hi, sums in rec["shards"]:
        for i, s in zip(range(lo, hi), sums):
            assert i not in got
            got[i] = s
    assert sorted(got) == list(range(6))
    for r in range(2):
        lo, hi = bench.shard_range(6, r, 2)
        pcm = bench.synth_batch(hi - lo, 16000, bench.rank_seed_base(r, 6, 2))
        for j, p in enumerate(pcm):
            assert got[lo + j] == float(np.float64(O.frontend_compute(p)).sum())
