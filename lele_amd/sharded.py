"""Utterances across GPUs.

Every utterance (and image) is a complete forward with no cross-item state, so rank r simply owns the block [lo, hi) of the
global batch and runs the single-GPU path on it.  The one exchange the recogniser has is at the very end: an all-gather of
the DECODED token ids (i32, a few hundred bytes per utterance -- greedy arg-max and the blank / special-token filter already
ran on the device, `lele_hip_argmax_last` + `lele_hip_token_filter`), so that every rank (or just rank 0) holds the
transcripts of the whole batch.  Logits never cross a link.

Two transports for the same packed rows:
  * `all_gather_ids_rccl` -- the product path: the C ABI's own communicator (`lele_hip_comm_*`, RCCL over xGMI, no torch).  The
    ids are packed ON THE DEVICE (count column + ids, -1 padding) and gathered on the ctx stream; only the gathered block is
    copied to the host, once.
  * `all_gather_ids` -- the same rows through `torch.distributed` (backend "gloo"), which is how the CPU tests exercise the
    N > 1 logic without a GPU."""
import numpy as np


def shard_range(total, rank, world):
    """static block partition of the global batch; rank r owns utterances [lo, hi) (sizes differ by at most one)"""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_ids(ids, counts, rows, width):
    """[rows, 1 + width] int32: column 0 = number of kept tokens, then the ids, -1 beyond them.  `ids` is the padded
    [n, W] matrix and `counts` the [n] vector `kernels.token_filter` returns; rows beyond n (ragged last shard) get count -1."""
    ids, counts = np.asarray(ids, np.int32), np.asarray(counts, np.int32).reshape(-1)
    if ids.ndim != 2 or ids.shape[0] != counts.shape[0] or ids.shape[0] > rows or ids.shape[1] > width:
        raise ValueError("pack_ids: ids %s / counts %s do not fit %d rows of width %d" % (ids.shape, counts.shape, rows, width))
    if counts.size and (counts.min() < 0 or counts.max() > ids.shape[1]):
        raise ValueError("pack_ids: a count lies outside [0, %d]" % ids.shape[1])
    out = np.full((rows, 1 + width), -1, np.int32)
    out[:ids.shape[0], 0] = counts
    out[:ids.shape[0], 1:1 + ids.shape[1]] = ids
    return out


def unpack_ids(packed):
    """rows of pack_ids -> list of int32 id arrays (padding rows dropped)"""
    return [row[1:1 + row[0]].copy() for row in np.asarray(packed) if row[0] >= 0]


def all_gather_ids(ids, counts, total, dist=None, device="cpu"):
    """Token ids of the whole batch, in global utterance order, on every rank.

    ids/counts: this rank's `token_filter` result (host arrays, shard_range(total, rank, world) utterances).  One MAX
    all-reduce agrees on the row width (shards may have different frame counts), one all-gather moves world x rows x (1+W)
    int32 -- C4: 32 x 172 x 4 B = 22 KB per GPU, far below any link's latency-bandwidth product, which is why it is a single
    flat collective and not bucketed."""
    ids = np.asarray(ids, np.int32)
    if dist is None:
        if ids.shape[0] != total:
            raise ValueError("all_gather_ids: %d utterances given, %d expected" % (ids.shape[0], total))
        return unpack_ids(pack_ids(ids, counts, total, ids.shape[1]))
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(total, rank, world)
    if ids.shape[0] != hi - lo:
        raise ValueError("all_gather_ids: rank %d owns %d utterances but was given %d" % (rank, hi - lo, ids.shape[0]))
    w = torch.tensor([ids.shape[1]], dtype=torch.int64, device=device)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    width, rows = int(w.item()), -(-total // world)
    mine = torch.from_numpy(pack_ids(ids, counts, rows, width)).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    out = []
    for r, p in enumerate(parts):
        got = unpack_ids(p.cpu().numpy())
        rlo, rhi = shard_range(total, r, world)
        if len(got) != rhi - rlo:
            raise RuntimeError("all_gather_ids: rank %d sent %d utterances, its shard has %d" % (r, len(got), rhi - rlo))
        out += got
    return out


def all_gather_ids_rccl(ids, counts, total, comm, ctx, bufs=None):
    """The exchange step on the device.  ids: device int32 [n, W] (kept ids first, -1 after), counts: device int32 [n] -- what
    `kernels.token_filter` returned on this rank for its `shard_range(total, rank, world)` utterances.  Rows are packed as
    [count | ids] on the device, padded to ceil(total / world) rows (ragged last shards) and to the widest shard (one MAX
    all-reduce of a scalar), gathered with ONE `lele_hip_comm_allgather_i32` on the ctx stream, and read back once.
    `bufs`: optional list of 4 LeleBufs to reuse across steps (no allocation in the steady state)."""
    from . import kernels as K
    rank, world = comm.rank, comm.world
    lo, hi = shard_range(total, rank, world)
    n, w = ids.shape
    if n != hi - lo:
        raise ValueError("all_gather_ids_rccl: rank %d owns %d utterances but was given %d" % (rank, hi - lo, n))
    b = bufs or [ctx.buf() for _ in range(4)]
    width = comm.allreduce_max(w) if world > 1 else w
    rows = -(-total // world)
    packed = K.concat([K.reshape(counts, [n, 1]), ids], 1, out=b[0], ctx=ctx)                 # [n, 1 + w]
    if width > w or rows > n:  # -1 fill: count -1 marks a padding row, id -1 a padding column
        packed = K.pad(packed, [0, 0, rows - n, width - w], np.array([-1], np.int32), "constant", out=b[1], ctx=ctx)
    flat = K.reshape(packed, [rows * (1 + width)])
    everything = comm.allgather_i32(flat.raw(), out=b[2]).numpy().reshape(world, rows, 1 + width)  # the one D2H copy
    out = []
    for r in range(world):
        got = unpack_ids(everything[r])
        rlo, rhi = shard_range(total, r, world)
        if len(got) != rhi - rlo:
            raise RuntimeError("all_gather_ids_rccl: rank %d sent %d utterances, its shard has %d" % (r, len(got), rhi - rlo))
        out += got
    return out


# ------------------------------------------------------------------------------------------------------------ images (configs[4])
# The reference returns `(logits [1, 300, 38], mask_features [1, 32, 160, 160])` per image (examples/yolo26n-seg/src/yolo26seg.rs:706-716)
# and its application reduces them to detections + a mask (image.rs:127-265).  Sharded over ranks, that reduction runs on the device
# per image (`lele_hip_yolo_seg_postprocess`), and what crosses a link is the compact result: one fixed-width row block per image,
# [count | 300 x 38 values, zeros behind the kept rows] -- 45.6 KB an image instead of the 3.3 MB of the raw pair.
DET_ROWS, DET_WIDTH = 300, 38


def pack_detections(dets, counts, rows):
    """[rows, 1 + 300 * 38] float32: column 0 = the image's number of kept detections (exact in f32), then its [300, 38] block with
    zeros behind the kept rows.  Images beyond len(counts) (ragged last shard) get count -1."""
    dets = np.asarray(dets, np.float32).reshape(-1, DET_ROWS, DET_WIDTH)
    counts = np.asarray(counts, np.int32).reshape(-1)
    if dets.shape[0] != counts.shape[0] or dets.shape[0] > rows:
        raise ValueError("pack_detections: %d images / %d counts do not fit %d rows" % (dets.shape[0], counts.shape[0], rows))
    if counts.size and (counts.min() < 0 or counts.max() > DET_ROWS):
        raise ValueError("pack_detections: a count lies outside [0, %d]" % DET_ROWS)
    out = np.zeros((rows, 1 + DET_ROWS * DET_WIDTH), np.float32)
    out[:, 0] = -1
    out[:counts.shape[0], 0] = counts
    keep = np.arange(DET_ROWS)[None, :] < counts[:, None]
    out[:counts.shape[0], 1:] = (dets * keep[:, :, None]).reshape(counts.shape[0], -1)
    return out


def unpack_detections(packed):
    """rows of pack_detections -> list of [count, 38] float32 arrays (padding rows dropped)"""
    out = []
    for row in np.asarray(packed, np.float32):
        if row[0] >= 0:
            out.append(row[1:].reshape(DET_ROWS, DET_WIDTH)[:int(row[0])].copy())
    return out


def _collect(parts, total, world, what):
    out = []
    for r, got in enumerate(parts):
        rlo, rhi = shard_range(total, r, world)
        if len(got) != rhi - rlo:
            raise RuntimeError("%s: rank %d sent %d images, its shard has %d" % (what, r, len(got), rhi - rlo))
        out += got
    return out


def all_gather_detections(dets, counts, total, dist=None, device="cpu"):
    """Detections of the whole batch, in global image order, on every rank -- `torch.distributed` form (gloo in the CPU tests)."""
    dets = np.asarray(dets, np.float32).reshape(-1, DET_ROWS, DET_WIDTH)
    if dist is None:
        if dets.shape[0] != total:
            raise ValueError("all_gather_detections: %d images given, %d expected" % (dets.shape[0], total))
        return unpack_detections(pack_detections(dets, counts, total))
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(total, rank, world)
    if dets.shape[0] != hi - lo:
        raise ValueError("all_gather_detections: rank %d owns %d images but was given %d" % (rank, hi - lo, dets.shape[0]))
    mine = torch.from_numpy(pack_detections(dets, counts, -(-total // world))).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return _collect([unpack_detections(p.cpu().numpy()) for p in parts], total, world, "all_gather_detections")


def all_gather_detections_rccl(dets, counts, total, comm, ctx, bufs=None):
    """The same exchange on the device: dets (device f32 [n, 300, 38], zeros behind the kept rows) and counts (device i32 [n]) are what
    `kernels.yolo_seg_postprocess` left on this rank for its `shard_range(total, rank, world)` images.  Two all-gathers on the ctx
    stream (`lele_hip_comm_allgather` for the rows, `lele_hip_comm_allgather_i32` for the counts), one read-back each.  A ragged last
    shard is padded with count -1 rows on the device."""
    from . import kernels as K
    rank, world = comm.rank, comm.world
    lo, hi = shard_range(total, rank, world)
    n = int(counts.shape[0])
    if n != hi - lo:
        raise ValueError("all_gather_detections_rccl: rank %d owns %d images but was given %d" % (rank, hi - lo, n))
    b = bufs or [ctx.buf() for _ in range(4)]
    rows = -(-total // world)
    d2 = K.reshape(dets, [n, DET_ROWS * DET_WIDTH])
    c1 = K.reshape(counts, [n])
    if rows > n:
        d2 = K.pad(d2, [0, 0, rows - n, 0], np.array([0], np.float32), "constant", out=b[0], ctx=ctx)
        c1 = K.pad(c1, [0, rows - n], np.array([-1], np.int32), "constant", out=b[1], ctx=ctx)
    all_d = comm.allgather(d2.raw(), out=b[2]).numpy().reshape(world, rows, DET_ROWS, DET_WIDTH)
    all_c = comm.allgather_i32(c1.raw(), out=b[3]).numpy().reshape(world, rows)
    parts = [[all_d[r, i, :all_c[r, i]].copy() for i in range(rows) if all_c[r, i] >= 0] for r in range(world)]
    return _collect(parts, total, world, "all_gather_detections_rccl")


# ------------------------------------------------------------------------------------------------------------ full logits (configs[3], on request)
# the preferred payload of configs[3] is the decoded token ids (684 B an utterance); "full logits if requested":
# f32 [T + 4, 25055] = 17.1 MB an utterance, 548 MB per GPU at 32 utterances -- one all-gather of the raw tensor, shards padded to
# ceil(total / world) utterances with zero rows that the receiver drops.
def all_gather_logits(logits, total, dist=None, device="cpu"):
    """logits f32 [n, T, V] of this rank's `shard_range(total, rank, world)` utterances -> [total, T, V] in global order on every rank
    (`torch.distributed` form: gloo in the CPU tests, nccl = RCCL with device tensors)"""
    logits = np.asarray(logits, np.float32)
    if dist is None:
        if logits.shape[0] != total:
            raise ValueError("all_gather_logits: %d utterances given, %d expected" % (logits.shape[0], total))
        return logits
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    lo, hi = shard_range(total, rank, world)
    if logits.shape[0] != hi - lo:
        raise ValueError("all_gather_logits: rank %d owns %d utterances but was given %d" % (rank, hi - lo, logits.shape[0]))
    rows = -(-total // world)
    mine = np.zeros((rows,) + logits.shape[1:], np.float32)
    mine[:hi - lo] = logits
    mine = torch.from_numpy(mine).to(device)
    parts = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return np.concatenate([p.cpu().numpy()[:shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r, p in enumerate(parts)], 0)


def all_gather_logits_rccl(logits, total, comm, ctx, bufs=None, to_host=True):
    """The same on the device through the C ABI: logits = device f32 [n, T, V]; ONE `lele_hip_comm_allgather` on the ctx stream (the
    tensor moves as bytes).  Returns the [total, T, V] host array (one read-back), or with to_host=False the gathered device tensor
    [world, rows, T, V] (rows = ceil(total / world); padding utterances are zero) for a consumer that stays on the device."""
    from . import kernels as K
    rank, world = comm.rank, comm.world
    lo, hi = shard_range(total, rank, world)
    n, t, v = (int(d) for d in logits.shape)
    if n != hi - lo:
        raise ValueError("all_gather_logits_rccl: rank %d owns %d utterances but was given %d" % (rank, hi - lo, n))
    b = bufs or [ctx.buf() for _ in range(2)]
    rows = -(-total // world)
    x = logits
    if rows > n:
        x = K.pad(x, [0, 0, 0, rows - n, 0, 0], np.array([0], np.float32), "constant", out=b[0], ctx=ctx)
    got = comm.allgather(x.raw(), out=b[1])
    if not to_host:
        return got
    every = got.numpy().reshape(world, rows, t, v)
    return np.concatenate([every[r, :shard_range(total, r, world)[1] - shard_range(total, r, world)[0]] for r in range(world)], 0)
