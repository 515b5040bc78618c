"""The exchange step of the sharded recogniser through the C ABI's own RCCL communicator (lele_hip_comm_*).

The GPU box has one device and RCCL refuses two ranks on one device, so what runs here is a 1-rank communicator -- RCCL is
loaded (dlopen), the unique id travels through a file, the collective is issued on the ctx stream from device memory -- plus the
native runner's `--ranks 1 --decode` loop.  The N > 1 logic (sharding, packing, ragged shards) is covered with gloo on CPU in
tests/test_multiprocess.py; N = 2..8 over xGMI is the driver's scaling run of bench.py."""
import json
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_communicator_through_the_c_abi(ctx, tmp_path):
    from lele_amd import kernels as K
    from lele_amd._lib import Comm, Weight
    from lele_amd.sharded import all_gather_ids, all_gather_ids_rccl
    comm = Comm.from_file(ctx, str(tmp_path / "uid"), 0, 1, timeout_ms=10000)
    assert (comm.rank, comm.world) == (0, 1)
    ids = np.arange(24, dtype=np.int32).reshape(4, 6) - 3
    got = comm.allgather_i32(ctx.buf().upload(ids.reshape(-1)))
    assert got.shape == (1, 24) and np.array_equal(got.numpy()[0], ids.reshape(-1))
    assert comm.allreduce_max(41) == 41
    comm.barrier()
    # the recogniser's tail: logits -> arg-max -> token filter on the device -> packed rows gathered -> transcripts on the host
    rng = np.random.default_rng(0)
    logits = rng.standard_normal((5, 37, 300)).astype(np.float32)
    skip = np.zeros(300, np.uint8)
    skip[0] = 1
    skip[250:] = 1
    kept, counts = K.token_filter(K.argmax_last(ctx.buf().upload(logits), ctx=ctx), Weight(skip), ctx=ctx)
    via_rccl = all_gather_ids_rccl(kept, counts, 5, comm, ctx)
    local = all_gather_ids(kept.numpy(), counts.numpy(), 5)
    assert len(via_rccl) == 5 and all(np.array_equal(a, b) for a, b in zip(via_rccl, local))
    am = logits.argmax(-1)
    for u in range(5):
        assert np.array_equal(via_rccl[u], np.array([t for t in am[u] if not skip[t]], np.int32))
    comm.close()


def test_gather_of_detection_rows_through_the_c_abi(ctx, tmp_path):
    """configs[4]'s exchange: post-processing on the device per image, then the fixed-width rows and their counts gathered with
    lele_hip_comm_allgather (any element type, moved as bytes) + lele_hip_comm_allgather_i32 -- a 1-rank communicator here (see the
    module docstring); the packing / ragged-shard logic at world 2 is tests/test_multiprocess.py's gloo test"""
    from lele_amd import kernels as K
    from lele_amd._lib import Comm
    from lele_amd.sharded import all_gather_detections, all_gather_detections_rccl
    from oracle import pyoracle as O
    from tests.test_app_steps import _seg_inputs
    comm = Comm.from_file(ctx, str(tmp_path / "uid"), 0, 1, timeout_ms=10000)
    x = np.arange(2 * 3 * 5, dtype=np.float32).reshape(2, 3, 5) / 7
    got = comm.allgather(ctx.buf().upload(x))
    assert got.shape == (1, 2, 3, 5) and got.dtype == np.float32 and np.array_equal(got.numpy()[0], x)
    u = np.arange(11, dtype=np.uint8)
    assert np.array_equal(comm.allgather(ctx.buf().upload(u)).numpy(), u[None])
    rng = np.random.default_rng(21)
    pairs = [_seg_inputs(rng, hm=32) for _ in range(3)]
    logits = np.concatenate([p[0].reshape(1, 300, 38) for p in pairs])
    feat = np.concatenate([p[1].reshape(1, 32, 32, 32) for p in pairs])
    dets, count, _mask = K.yolo_seg_postprocess(logits, feat, 64, 64, 0.5, 80, ctx=ctx)
    via_rccl = all_gather_detections_rccl(dets, count, 3, comm, ctx)
    local = all_gather_detections(dets.numpy(), count.numpy(), 3)
    assert len(via_rccl) == 3 and all(np.array_equal(a, b) for a, b in zip(via_rccl, local))
    for i, (lg, ft) in enumerate(pairs):
        assert np.array_equal(via_rccl[i], O.yolo_seg_postprocess(lg, ft, 64, 64, 0.5)[0])
    comm.close()


def test_gather_of_full_logits_through_the_c_abi(ctx, tmp_path):
    """section 8(e)'s other payload: the raw logits (17.1 MB an utterance at full size) with ONE lele_hip_comm_allgather -- a 1-rank
    communicator here; world 2 on this one GPU runs through bench.py --gather-logits (test_bench_two_ranks_on_one_gpu_end_to_end)"""
    from lele_amd._lib import Comm
    from lele_amd.sharded import all_gather_logits_rccl
    comm = Comm.from_file(ctx, str(tmp_path / "uid"), 0, 1, timeout_ms=10000)
    x = np.random.default_rng(4).standard_normal((3, 175, 2505)).astype(np.float32)
    from lele_amd.tensor import TensorView
    got = all_gather_logits_rccl(TensorView(ctx.buf().upload(x)), 3, comm, ctx)
    assert got.shape == x.shape and np.array_equal(got, x)
    dev = all_gather_logits_rccl(TensorView(ctx.buf().upload(x)), 3, comm, ctx, to_host=False)
    assert tuple(dev.shape)[0] == 1 and np.array_equal(dev.numpy().reshape(x.shape), x)
    comm.close()


def test_rendezvous_file_of_another_job_is_refused(ctx, tmp_path, monkeypatch):
    """[32-byte job token][128-byte id]: a reader takes only a file that carries the digest of ITS LELE_JOB_ID -- not what an earlier
    job left under the same name (whatever its age), not a file of another size -- and rank 0 replaces what it finds"""
    from lele_amd import _lib
    from lele_amd._lib import Comm
    path = tmp_path / "uid"
    monkeypatch.setenv("LELE_JOB_ID", "job-b")
    for stale in (b"\0" * 160, b"\1" * 128, b"\7" * 161):
        path.write_bytes(stale)
        with pytest.raises(_lib.LeleError, match="waited"):
            Comm.from_file(ctx, str(path), 1, 2, timeout_ms=40)
    comm = Comm.from_file(ctx, str(path), 0, 1, timeout_ms=2000)
    data = path.read_bytes()
    assert len(data) == 160 and data[:32] != b"\0" * 32 and data != b"\0" * 160
    comm.close()
    monkeypatch.setenv("LELE_JOB_ID", "job-c")     # the next job does not take job-b's file either
    with pytest.raises(_lib.LeleError, match="waited"):
        Comm.from_file(ctx, str(path), 1, 2, timeout_ms=40)
    # no token at all: a well-formed file that nobody keeps BEATING (what a dead token-less job left) is refused whatever its age --
    # a minute old or written this instant (no wall-clock test; tests/test_multiprocess.py has the live handshake)
    monkeypatch.delenv("LELE_JOB_ID")
    monkeypatch.delenv("TORCHELASTIC_RUN_ID", raising=False)
    for stale in (b"\0" * 32 + b"\5" * 128, b"\0" * 32 + b"\5" * 128 + b"\3" + b"\0" * 7):
        path.write_bytes(stale)
        with pytest.raises(_lib.LeleError, match="waited"):
            Comm.from_file(ctx, str(path), 1, 2, timeout_ms=40)
        old = os.stat(path).st_mtime - 60
        os.utime(path, (old, old))
        with pytest.raises(_lib.LeleError, match="waited"):
            Comm.from_file(ctx, str(path), 1, 2, timeout_ms=40)
    comm = Comm.from_file(ctx, str(path), 0, 1, timeout_ms=2000)   # a token-less rank 0 alone: [zero token][id][beat]
    assert len(path.read_bytes()) == 168
    comm.close()


def test_native_runner_ranks_and_decode(ctx, tmp_path):
    """lele_run --ranks 1 --decode: fork, file rendezvous, plan, arg-max on the device, RCCL all-gather, rank 0 prints"""
    torch = pytest.importorskip("torch")
    from lele_amd.compiler import compile_model
    from tests.onnx_util import export
    from tests.test_compiler import Toy
    x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(3))
    plan, blob = compile_model(export(Toy(), (x,), opset=13))
    (tmp_path / "p.json").write_text(json.dumps(plan))
    (tmp_path / "w.bin").write_bytes(blob)
    x.numpy().tofile(tmp_path / "x.bin")
    exe = os.path.join(ROOT, "lele_amd", "lele_run")
    r = subprocess.run([exe, str(tmp_path / "p.json"), str(tmp_path / "w.bin"), "--input", "x=%s:f32:2,3,16,16" % (tmp_path / "x.bin"), "--out",
                        str(tmp_path / "o"), "--ranks", "1", "--decode"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    out = np.fromfile(tmp_path / "o0.bin", np.float32).reshape(rec["outputs"][0])
    assert rec["ranks"] == 1 and rec["gathered_ids"] == [1, int(np.prod(out.shape[:-1]))]
    # tokenizer.rs:50-61: the LAST of equal maxima wins
    last_max = out.shape[-1] - 1 - np.flip(out, -1).argmax(-1)
    assert rec["ids_checksum"] == int(last_max.sum())


def test_bench_two_ranks_on_one_gpu_end_to_end():
    """`bench.py --gpus 2` starts its own two ranks.  On a one-GPU box both share the device (LELE_BENCH_SHARE_GPU) and the process
    group is gloo; RCCL refuses two ranks on one device, so the C ABI communicator cannot come up -- every rank must then agree
    to move the ids through the process group instead (bench.py permits that only under --allow-fallback), and the line must say so.  What this pins on real hardware: the N > 1
    control flow (sharding by rank, fences, MAX over ranks, the transport agreement, the gather and its self-check) and that the
    JSON line is the LAST line on stdout whatever the collective library prints."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, LELE_BENCH_SHARE_GPU="1", LELE_BENCH_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64",
                        "--layers", "2", "--per-gpu", "4", "--sv-steps", "2", "--allow-fallback", "--gather-logits"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    last = r.stdout.strip().splitlines()[-1]
    line = json.loads(last)
    assert line["n_gpus"] == 2 and line["config"]["batch_per_gpu"] == 64
    sv = line["sensevoice"]
    assert sv["c4_utterances"] == 8 and sv["c4_gathered_ok"] is True
    assert "rccl" in sv["c4_collective"] or "torch.distributed" in sv["c4_collective"]
    lg = sv["c4_logits_gather"]                                            # the full-logits payload at world 2 on this one GPU
    assert lg["utterances"] == 8 and lg["own_block_equals_local_logits"] is True and lg["bytes_per_rank"] == 4 * lg["tokens"] * 25055 * 4
    assert "rtf_model" not in line and "cpu_baseline" not in line          # N = 1 only
    g = line["yolo"]["gather"]                                             # configs[4]: every image's detections reached every rank
    assert g["images"] == g["images_expected"] == 2 * 64 and g["own_block_equals_local_postprocess"] is True
    assert "rccl" in g["collective"] or "torch.distributed" in g["collective"]
