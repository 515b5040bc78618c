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


GATHER_WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["LELE_ROOT"])
import torch.distributed as dist
from lele_amd.sharded import all_gather_ids, shard_range
from oracle import pyoracle as O

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total, vocab = 5, 40
skip = np.zeros(vocab, np.uint8); skip[[0, 1, 2]] = 1
lo, hi = shard_range(total, rank, world)
frames = 9 + 4 * rank                      # shards with different frame counts: the row width has to be agreed on
logits = np.stack([np.random.default_rng(100 + i).standard_normal((frames, vocab)).astype(np.float32) for i in range(lo, hi)])
ids, counts = O.decode_greedy_ids(logits, skip)   # stands in for argmax_last + token_filter on the device
everything = all_gather_ids(ids, counts, total, dist)
print(json.dumps({"rank": rank, "ids": [[int(v) for v in a] for a in everything]}))
dist.destroy_process_group()
'''


def test_all_gather_of_decoded_ids_two_ranks_gloo(tmp_path):
    """the recogniser's only exchange (section 8e): every rank ends up with the token ids of the whole batch, in order"""
    from lele_amd.sharded import shard_range
    from oracle import pyoracle as O
    port = _free_port()
    script = tmp_path / "gather_worker.py"
    script.write_text(GATHER_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LELE_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    skip = np.zeros(40, np.uint8)
    skip[[0, 1, 2]] = 1
    want = []
    for r in range(2):
        lo, hi = shard_range(5, r, 2)
        for i in range(lo, hi):
            lg = np.random.default_rng(100 + i).standard_normal((1, 9 + 4 * r, 40)).astype(np.float32)
            ids, counts = O.decode_greedy_ids(lg, skip)
            want.append([int(v) for v in ids[0, :counts[0]]])
    assert any(len(w) > 9 for w in want[3:])          # the wider shard really is wider than the narrow one's row
    for out, _err in outs:
        assert json.loads(out.strip().splitlines()[-1])["ids"] == want


DET_WORKER = '''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["LELE_ROOT"])
import torch.distributed as dist
from lele_amd.sharded import all_gather_detections, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total = 5
lo, hi = shard_range(total, rank, world)
dets = np.stack([np.random.default_rng(300 + i).standard_normal((300, 38)).astype(np.float32) for i in range(lo, hi)])
counts = np.array([(7 * i + 3) % 301 for i in range(lo, hi)], np.int32)   # what yolo_seg_postprocess leaves: `count` kept rows per image
everything = all_gather_detections(dets, counts, total, dist)
print(json.dumps({"rank": rank, "counts": [int(d.shape[0]) for d in everything], "sums": [float(d.astype(np.float64).sum()) for d in everything]}))
dist.destroy_process_group()
'''


def test_all_gather_of_detections_two_ranks_gloo(tmp_path):
    """configs[4]'s only exchange (section 8e, "C5"): every rank ends up with the kept detections of every image of the batch, in global
    image order -- 5 images over 2 ranks (a ragged last shard), fixed-width rows, the garbage behind an image's kept rows never sent"""
    port = _free_port()
    script = tmp_path / "det_worker.py"
    script.write_text(DET_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LELE_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    want_counts = [(7 * i + 3) % 301 for i in range(5)]
    want_sums = [float(np.random.default_rng(300 + i).standard_normal((300, 38)).astype(np.float32)[:want_counts[i]].astype(np.float64).sum()) for i in range(5)]
    for out, _err in outs:
        rec = json.loads(out.strip().splitlines()[-1])
        assert rec["counts"] == want_counts and np.allclose(rec["sums"], want_sums, rtol=0, atol=1e-9)


LOGITS_WORKER = '''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["LELE_ROOT"])
import torch.distributed as dist
from lele_amd.sharded import all_gather_logits, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
total = 5
lo, hi = shard_range(total, rank, world)
mine = np.stack([np.random.default_rng(700 + i).standard_normal((21, 64)).astype(np.float32) for i in range(lo, hi)])
every = all_gather_logits(mine, total, dist)
print(json.dumps({"rank": rank, "shape": list(every.shape), "sums": [float(u.astype(np.float64).sum()) for u in every]}))
dist.destroy_process_group()
'''


def test_all_gather_of_full_logits_two_ranks_gloo(tmp_path):
    """section 8(e), "full logits if requested": the raw [T, V] tensors of every utterance on every rank, in global order, a ragged
    last shard (5 utterances over 2 ranks: 3 + 2) padded for the collective and trimmed behind it"""
    port = _free_port()
    script = tmp_path / "logits_worker.py"
    script.write_text(LOGITS_WORKER)
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LELE_ROOT=ROOT)
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    want = [float(np.random.default_rng(700 + i).standard_normal((21, 64)).astype(np.float32).astype(np.float64).sum()) for i in range(5)]
    for out, _err in outs:
        rec = json.loads(out.strip().splitlines()[-1])
        assert rec["shape"] == [5, 21, 64] and rec["sums"] == want


def test_pack_detections_round_trip_and_refusals():
    import pytest
    from lele_amd.sharded import all_gather_detections, pack_detections, unpack_detections
    dets = np.random.default_rng(1).standard_normal((2, 300, 38)).astype(np.float32)
    counts = np.array([300, 0], np.int32)
    packed = pack_detections(dets, counts, 3)
    assert packed.shape == (3, 1 + 300 * 38) and packed[:, 0].tolist() == [300.0, 0.0, -1.0] and not packed[1:, 1:].any()
    got = unpack_detections(packed)
    assert len(got) == 2 and np.array_equal(got[0], dets[0]) and got[1].shape == (0, 38)
    assert [g.shape[0] for g in all_gather_detections(dets, counts, 2)] == [300, 0]
    with pytest.raises(ValueError):
        pack_detections(dets, counts, 1)
    with pytest.raises(ValueError):
        pack_detections(dets, np.array([301, 0], np.int32), 3)
    with pytest.raises(ValueError):
        all_gather_detections(dets, counts, 3)


def test_pack_ids_rejects_what_does_not_fit():
    import pytest
    from lele_amd.sharded import all_gather_ids, pack_ids, unpack_ids
    ids, counts = np.array([[4, 5, -1], [6, -1, -1]], np.int32), np.array([2, 1], np.int32)
    packed = pack_ids(ids, counts, 3, 4)
    assert packed.tolist() == [[2, 4, 5, -1, -1], [1, 6, -1, -1, -1], [-1, -1, -1, -1, -1]]
    assert [a.tolist() for a in unpack_ids(packed)] == [[4, 5], [6]]
    assert [a.tolist() for a in all_gather_ids(ids, counts, 2)] == [[4, 5], [6]]
    with pytest.raises(ValueError):
        pack_ids(ids, counts, 1, 4)
    with pytest.raises(ValueError):
        pack_ids(ids, np.array([2, 7], np.int32), 3, 4)
    with pytest.raises(ValueError):
        all_gather_ids(ids, counts, 3)


def test_bench_spawns_its_own_ranks_and_refuses_a_mismatched_world():
    """`python bench.py --gpus N` with no launcher environment starts N ranks itself and rank 0 prints n_gpus = N; under a
    launcher whose WORLD_SIZE differs from --gpus it refuses to run (--gpus was parsed and never used).  --dry-run
    keeps the launcher / rendezvous / fence / MAX-over-ranks logic and skips the GPU work."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    bench_py = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py")
    out = subprocess.run([sys.executable, bench_py, "--gpus", "3", "--steps", "7", "--warmup", "2", "--dry-run"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout          # exactly ONE line, from rank 0
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 3 and rec["steps"] == 7 and rec["warmup"] == 2
    assert abs(rec["value"] - 0.003) < 1e-9      # MAX over the three ranks' (rank + 1) ms
    bad = subprocess.run([sys.executable, bench_py, "--gpus", "4", "--dry-run"], env=dict(env, WORLD_SIZE="2", RANK="0"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert bad.returncode != 0 and "refusing" in bad.stderr


def test_tokenless_rendezvous_file_needs_a_live_writer(tmp_path, monkeypatch):
    """lele_hip_comm_read_id_file (the reader's half of comm_init_file; no device): without LELE_JOB_ID / TORCHELASTIC_RUN_ID a
    file is accepted once it has been seen under two different beats -- however long before the reader it was first written
    (start-up skew of many seconds, file-server clocks) -- and never when nobody keeps it beating."""
    import ctypes as C
    import threading
    import time
    from lele_amd import _lib
    fn = _lib.lib().lele_hip_comm_read_id_file
    fn.restype = C.c_int
    fn.argtypes = [C.c_char_p, C.c_int, C.c_char_p]
    monkeypatch.delenv("LELE_JOB_ID", raising=False)
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    path = tmp_path / "uid"
    ident = bytes(range(128))

    def write(beat):
        tmp = tmp_path / "uid.tmp"
        tmp.write_bytes(b"\0" * 32 + ident + int(beat).to_bytes(8, "little"))
        os.replace(tmp, path)

    got = C.create_string_buffer(128)
    write(7)
    old = os.stat(path).st_mtime - 3600
    os.utime(path, (old, old))
    assert fn(str(path).encode(), 60, got) != 0                     # an hour old, no writer: refused
    write(7)
    assert fn(str(path).encode(), 60, got) != 0                     # brand new, no writer: refused all the same
    stop = threading.Event()

    def rank0():
        beat = 8
        while not stop.is_set():
            time.sleep(0.02)
            write(beat)
            beat += 1

    write(7)
    os.utime(path, (old, old))                                       # "written long before the reader arrived"
    t = threading.Thread(target=rank0)
    t.start()
    try:
        assert fn(str(path).encode(), 5000, got) == 0
    finally:
        stop.set()
        t.join()
    assert got.raw == ident
    # with a token the file is [token][id] and only the token counts (age, beats: irrelevant); another job's token is refused
    monkeypatch.setenv("LELE_JOB_ID", "job-x")
    assert fn(str(path).encode(), 40, got) != 0


W8_WORKER = r'''
import json, os, sys
import numpy as np
sys.path.insert(0, os.environ["LELE_ROOT"])
import torch.distributed as dist
from lele_amd.sharded import all_gather_detections, all_gather_ids, shard_range
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
# configs[3]: 250 utterances (ragged: 250 = 8 x 31 + 2), rows of DIFFERENT widths per rank (10 s utterances decode to different lengths)
total_u = 250
lo, hi = shard_range(total_u, rank, world)
width = 20 + 3 * rank
ids = np.full((hi - lo, width), -1, np.int32)
counts = np.array([(5 * i + 1) % (width + 1) for i in range(lo, hi)], np.int32)
for j, i in enumerate(range(lo, hi)):
    ids[j, :counts[j]] = (np.arange(counts[j]) * 7 + i) % 25055
every_u = all_gather_ids(ids, counts, total_u, dist)
# configs[4]: 61 images (ragged: 8 x 7 + 5)
total_i = 61
lo2, hi2 = shard_range(total_i, rank, world)
dets = np.stack([np.full((300, 38), float(i), np.float32) for i in range(lo2, hi2)]) if hi2 > lo2 else np.zeros((0, 300, 38), np.float32)
cnt = np.array([(11 * i + 2) % 301 for i in range(lo2, hi2)], np.int32)
every_i = all_gather_detections(dets, cnt, total_i, dist)
ok_u = len(every_u) == total_u and all(len(a) == (5 * i + 1) % (20 + 3 * r + 1) and (len(a) == 0 or int(a[0]) == i % 25055)
                                         for r in range(world) for i, a in ((i, every_u[i]) for i in range(*shard_range(total_u, r, world))))
ok_i = len(every_i) == total_i and all(d.shape == ((11 * i + 2) % 301, 38) and (d.size == 0 or float(d[0, 0]) == float(i)) for i, d in enumerate(every_i))
print(json.dumps({"rank": rank, "ok_u": bool(ok_u), "ok_i": bool(ok_i), "shard_u": [lo, hi], "shard_i": [lo2, hi2]}))
dist.destroy_process_group()
'''


def test_eight_rank_gloo_gathers_with_ragged_shards(tmp_path):
    """nobody has run N = 8 on hardware, so everything short of RCCL itself runs here at world 8: the block
    partition of 250 utterances and 61 images (both ragged), the id gather with a different row width on every rank and the detection
    gather -- every rank must end up with every unit, in global order, exactly once."""
    port = _free_port()
    script = tmp_path / "w8.py"
    script.write_text(W8_WORKER)
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LELE_ROOT=ROOT,
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), [o[1][-500:] for o in outs]
    recs = sorted((json.loads(o.strip().splitlines()[-1]) for o, _e in outs), key=lambda r: r["rank"])
    assert all(r["ok_u"] and r["ok_i"] for r in recs), recs
    assert [r["shard_u"] for r in recs][0][0] == 0 and recs[-1]["shard_u"][1] == 250 and recs[-1]["shard_i"][1] == 61
    assert all(a["shard_u"][1] == b["shard_u"][0] and a["shard_i"][1] == b["shard_i"][0] for a, b in zip(recs, recs[1:]))


def test_bench_dry_run_at_eight_ranks_one_device_per_rank():
    """`python bench.py --gpus 8 --dry-run`: eight self-spawned ranks, rank 0's single line says 8 ranks were seen, every rank would open
    its own device (LOCAL_RANK 0..7), the MAX over ranks is rank 7's time, and the shards of both sharded legs tile the batch"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run"], env=dict(env, OMP_NUM_THREADS="1"),
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 8 and rec["ranks_seen"] == 8 and rec["one_device_per_rank"] and rec["devices"] == list(range(8))
    assert abs(rec["value"] - 0.008) < 1e-9
    u, im = rec["utterance_shards"], rec["image_shards"]
    assert u[0][0] == 0 and u[-1][1] == 256 and all(a[1] == b[0] and a[1] - a[0] == 32 for a, b in zip(u, u[1:]))
    assert im[0][0] == 0 and im[-1][1] == 512 and all(a[1] == b[0] for a, b in zip(im, im[1:]))
