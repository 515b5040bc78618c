"""Data-movement operators (bit-exact class): numpy oracle pinned on the reference's KATs (CPU); device == oracle (GPU)."""
import numpy as np
import pytest

from oracle import npref


def test_numpy_oracle_kats():
    # tests/kernel_accuracy.rs:152-171 concat, 193-203 expand, 272-300 split, 302-338 transpose, 355-374 gather
    assert np.array_equal(np.concatenate([np.array([[1, 2], [3, 4]], np.float32), np.array([[5, 6]], np.float32)], 0).ravel(),
                          [1, 2, 3, 4, 5, 6])
    assert np.array_equal(npref.gather(np.arange(1, 10, dtype=np.float32).reshape(3, 3), [0.0, 2.0], 0),
                          [[1, 2, 3], [7, 8, 9]])
    # tests/regression_kernels.rs:389-418 pad constant
    x = np.array([1, 2, 3, 4, 5], np.float32).reshape(1, 1, 1, 5)
    assert np.array_equal(npref.pad(x, [0, 0, 0, 2, 0, 0, 0, 2]).ravel(), [0, 0, 1, 2, 3, 4, 5, 0, 0])
    x = np.array([1, 2, 3, 4], np.float32).reshape(1, 1, 2, 2)
    assert np.array_equal(npref.pad(x, [0, 0, 1, 1, 0, 0, 1, 1], 99.0).ravel(),
                          [99, 99, 99, 99, 99, 1, 2, 99, 99, 3, 4, 99, 99, 99, 99, 99])
    # tests/regression_kernels.rs:364-387 reflect keeps the centre
    x = np.arange(1, 7, dtype=np.float32).reshape(1, 1, 2, 3)
    r = npref.pad(x, [0, 0, 1, 1, 0, 0, 1, 1], mode="reflect")
    assert r.shape == (1, 1, 4, 5) and np.array_equal(r[0, 0, 1:3, 1:4], x[0, 0])
    # src/kernels/conv2d.rs:3390-3749 resize / maxpool unit tests: 2x nearest upsampling repeats pixels
    x = np.arange(4, dtype=np.float32).reshape(1, 1, 2, 2)
    assert np.array_equal(npref.resize_nearest(x, 4, 4)[0, 0], np.repeat(np.repeat(x[0, 0], 2, 0), 2, 1))
    # tests/regression_kernels.rs:258-358 maxpool vs the in-test scalar oracle (here: brute force)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 2, 7, 9)).astype(np.float32)
    got = npref.max_pool2d(x, [3, 3], [2, 2], [1, 1, 1, 1])
    for oh in range(got.shape[2]):
        for ow in range(got.shape[3]):
            win = x[0, 0, max(0, oh * 2 - 1):oh * 2 + 2, max(0, ow * 2 - 1):ow * 2 + 2]
            assert got[0, 0, oh, ow] == win.max()
    v, i = npref.topk(np.array([[1, 3, 2, 3]], np.float32), 3)
    assert np.array_equal(v, [[3, 3, 2]]) and np.array_equal(i, [[1, 3, 2]])  # stable: first 3 wins the tie


def test_view_ops_host_only():
    # shape.rs:186-223 unit tests (reshape / flatten / squeeze / unsqueeze): no device needed, views move no data
    from lele_amd import kernels as Kk
    t = np.array([1, 2, 3, 4], np.float32).reshape(2, 2)
    assert Kk.reshape(t, [4]).shape == (4,) and Kk.reshape(t, [1, -1]).shape == (1, 4)
    assert Kk.reshape(np.zeros((2, 3, 4), np.float32), [0, -1]).shape == (2, 12)
    big = np.zeros((2, 3, 4), np.float32)
    assert Kk.flatten(big, 1).shape == (2, 12) and Kk.flatten(big, 2).shape == (6, 4)
    one = np.ones((1, 1), np.float32)
    assert Kk.squeeze(one).shape == () and Kk.unsqueeze(Kk.squeeze(one), [0]).shape == (1,)
    assert np.array_equal(Kk.shape(big).numpy(), [2, 3, 4]) and int(Kk.size(big).numpy()) == 24
    import lele_amd
    with pytest.raises(lele_amd.LeleError, match="element count mismatch"):
        Kk.reshape(t, [3])


@pytest.mark.gpu
def test_device_slice_transpose_expand_tile_split_concat(ctx):
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(1)
    x = rng.standard_normal((4, 5, 6)).astype(np.float32)
    big = 2 ** 63 - 1
    cases = [([1], [3], [0], []), ([0, 1], [4, 4], [0, 2], [1, 2]), ([-1], [-big - 1], [1], [-1]), ([2], [big], [2], [1]),
             ([-3], [100], [2], []), ([5], [1], [1], [-2]), ([1, 0, 2], [3, 5, 6], [], []), ([3], [1], [0], [])]
    for st, en, ax, sp in cases:
        got = Kk.slice(x, st, en, ax, sp, ctx=ctx)
        ref = npref.slice_(x, st, en, ax, sp)
        assert got.shape == ref.shape and np.array_equal(got.numpy(), ref), (st, en, ax, sp)
    for perm in ([], [0, 2, 1], [2, 0, 1], [1, 0, 2]):
        assert np.array_equal(Kk.transpose(x, perm, ctx=ctx).numpy(), np.transpose(x, perm or None))
    x4 = rng.standard_normal((1, 504, 4, 128)).astype(np.float32)  # the per-layer head split of SenseVoice
    assert np.array_equal(Kk.transpose(x4, [0, 2, 1, 3], ctx=ctx).numpy(), np.transpose(x4, [0, 2, 1, 3]))
    xi = rng.integers(-9, 9, (3, 4)).astype(np.int64)
    assert np.array_equal(Kk.transpose(xi, [1, 0], ctx=ctx).numpy(), xi.T)
    e = np.array([[1.0], [2.0], [3.0]], np.float32)
    assert np.array_equal(Kk.expand(e, [3, 4], ctx=ctx).numpy(), np.broadcast_to(e, (3, 4)))  # kernel_accuracy.rs:193-203
    assert np.array_equal(Kk.expand(e, [2, 0, 2], ctx=ctx).numpy(), np.broadcast_to(e, (2, 3, 2)))
    assert np.array_equal(Kk.tile(x[:2, :2, :3], [2, 1, 3], ctx=ctx).numpy(), np.tile(x[:2, :2, :3], [2, 1, 3]))
    parts = Kk.split(x, 1, [2, 3], ctx=ctx)
    assert np.array_equal(parts[0].numpy(), x[:, :2]) and np.array_equal(parts[1].numpy(), x[:, 2:])
    c = Kk.concat([x[:, :2], x[:, 2:], x[:, :1]], 1, ctx=ctx)
    assert np.array_equal(c.numpy(), np.concatenate([x, x[:, :1]], 1))
    c = Kk.concat([parts[0], parts[1]], -2, ctx=ctx)  # device-resident inputs
    assert np.array_equal(c.numpy(), x)


@pytest.mark.gpu
def test_device_pad_gather_resize_pool_topk(ctx):
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 2, 5, 7)).astype(np.float32)
    for pads, mode, cv in (([0, 0, 1, 2, 0, 0, 3, 1], "constant", None), ([0, 0, 1, 2, 0, 0, 3, 1], "constant", [99.0]),
                           ([2, 1, 2, 1], "edge", None), ([0, 0, 2, 3, 0, 0, 1, 2], "reflect", None),
                           ([0, 0, 4, 6, 0, 0, 4, 6], "reflect", None)):
        got = Kk.pad(x, pads, cv, mode, ctx=ctx)
        ref = npref.pad(x, pads, 0 if cv is None else cv[0], mode)
        assert got.shape == ref.shape and np.array_equal(got.numpy(), ref), (pads, mode)
    sig = rng.standard_normal((1, 576)).astype(np.float32)  # Silero's reflect pad of the chunk
    assert np.array_equal(Kk.pad(sig, [0, 64, 0, 64], None, "reflect", ctx=ctx).numpy(),
                          npref.pad(sig, [0, 64, 0, 64], mode="reflect"))
    emb = rng.standard_normal((16, 8)).astype(np.float32)
    for idx in (np.array([0.0, 2.0], np.float32), np.array([[3, -1], [0, 15]], np.int64), np.array([5], np.int32)):
        assert np.array_equal(Kk.gather(emb, idx, 0, ctx=ctx).numpy(), npref.gather(emb, np.where(idx < 0, idx + 16, idx), 0))
    assert np.array_equal(Kk.gather(emb, np.array([1.0, 7.0], np.float32), 1, ctx=ctx).numpy(), emb[:, [1, 7]])
    gi = rng.integers(-5, 5, (4, 8)).astype(np.float32)
    assert np.array_equal(Kk.gather_elements(emb[:5], gi, 0, ctx=ctx).numpy(), npref.gather_elements(emb[:5], gi, 0))
    f = rng.standard_normal((2, 3, 20, 20)).astype(np.float32)
    for kw in (dict(scales=[1, 1, 2, 2]), dict(scales=[1, 1, 4, 4]), dict(scales=[1, 1, 8, 8]), dict(scales=[1, 1, 4, 2]),
               dict(sizes=[2, 3, 33, 17]), dict(scales=[1, 1, 0.5, 1.5])):
        for mode in ("asymmetric", "half_pixel"):
            got = Kk.resize_nearest(f, coordinate_transform_mode=mode, ctx=ctx, **kw)
            oh, ow = got.shape[2:]
            assert np.array_equal(got.numpy(), npref.resize_nearest(f, oh, ow, mode == "asymmetric")), (kw, mode)
    for args in (([5, 5], [1, 1], [2, 2, 2, 2]), ([2, 2], [2, 2], []), ([3, 3], [2, 2], [1, 1, 1, 1]), ([3], [2], [0], [2])):
        got = Kk.max_pool2d(f, *args, ctx=ctx)
        ref = npref.max_pool2d(f, *args)
        assert got.shape == ref.shape and np.array_equal(got.numpy(), ref), args
    assert np.array_equal(Kk.max_pool2d(f, [3, 3], [2, 2], [0, 0, 0, 0], [], True, ctx=ctx).numpy(),
                          npref.max_pool2d(f, [3, 3], [2, 2], [0, 0, 0, 0], [], True))
    s = rng.standard_normal((3, 300)).astype(np.float32)
    s[0, 10] = s[0, 20] = 9.0  # tie: the lower index must come first
    for largest in (True, False):
        v, i = Kk.topk(s, 17, largest=largest, ctx=ctx)
        rv, ri = npref.topk(s, 17, largest)
        assert np.array_equal(v.numpy(), rv) and np.array_equal(i.numpy(), ri)
    v, i = Kk.topk(s[:, :5], 50, ctx=ctx)
    assert v.shape == (3, 5)


@pytest.mark.gpu
def test_device_range_fill_cast(ctx):
    from lele_amd import kernels as Kk
    r = Kk.range([0.5], [4.2], [0.7], ctx=ctx).numpy()
    assert np.array_equal(r, (np.float32(0.5) + np.arange(6, dtype=np.float32) * np.float32(0.7)).astype(np.float32))
    assert Kk.range([3.0], [1.0], [1.0], ctx=ctx).shape == (0,)
    c = Kk.constant_of_shape(np.array([2, 3], np.int64), 1.5, ctx=ctx)
    assert c.shape == (2, 3) and np.all(c.numpy() == 1.5)
    ci = Kk.constant_of_shape(np.array([4], np.int64), 7, np.int64, ctx=ctx)
    assert ci.dtype == np.int64 and np.array_equal(ci.numpy(), [7, 7, 7, 7])
    x = np.array([1.7, -2.2, 3.0], np.float32)
    assert np.array_equal(Kk.cast_to_i64(x, ctx=ctx).numpy(), x.astype(np.int64))
    assert np.array_equal(Kk.cast_to_f32(np.array([5, -6], np.int64), ctx=ctx).numpy(), [5.0, -6.0])
    # TensorView::reinterpret_as_u8 (src/tensor.rs:92-97): Rust's `x as u8` = truncate toward zero, saturate, NaN -> 0
    codes = np.array([0.0, 1.0, 254.9, 255.0, 256.0, 1e9, -0.9, -3.0, 127.5, np.nan, np.inf, -np.inf], np.float32)
    want = np.array([0, 1, 254, 255, 255, 255, 0, 0, 127, 0, 255, 0], np.uint8)
    assert np.array_equal(Kk.reinterpret_as_u8(codes, ctx=ctx).numpy(), want)
    u = np.arange(256, dtype=np.uint8)
    assert np.array_equal(Kk.cast_to_i64(u, ctx=ctx).numpy(), u.astype(np.int64))
    assert np.array_equal(Kk.cast_to_i64(u.view(np.int8), ctx=ctx).numpy(), u.view(np.int8).astype(np.int64))


@pytest.mark.gpu
def test_device_topk_long_rows_with_ties(ctx):
    # rows longer than the 2048-element LDS chunk, heavy ties (stable: the lower index ranks first), both directions
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(7)
    # (8400 / 24000: the Yolo26n-seg tail; 28672 is the longest row whose keys are staged in LDS, 28673 the first that is swept from L2)
    for n, k in ((5000, 300), (24000, 300), (8400, 300), (28672, 300), (28673, 301), (40000, 1024), (2049, 2049), (300, 7), (80, 1), (81, 5), (3, 3)):
        x = np.round(rng.standard_normal((3, n)) * 3).astype(np.float32)
        if n >= 5000:  # signed zeros tie with each other, a NaN ranks below everything
            x[0, ::7] = -0.0
            x[0, 3::11] = 0.0
            x[1, 5::13] = np.nan
        for largest in (True, False):
            v, i = Kk.topk(x, k, -1, largest, True, ctx=ctx)
            rv, ri = npref.topk(x, k, largest)
            assert np.array_equal(v.numpy(), rv) and np.array_equal(i.numpy(), ri.astype(np.float32))


def test_adaptive_avg_pool1d_oracle():
    # pooling.rs:1-30: windows [floor(i*L/O), ceil((i+1)*L/O)); equal split, overlapping windows, up-sampling, O = 0
    x = np.arange(12, dtype=np.float32).reshape(2, 6)
    assert np.array_equal(npref.adaptive_avg_pool1d(x, 3), [[0.5, 2.5, 4.5], [6.5, 8.5, 10.5]])
    assert np.array_equal(npref.adaptive_avg_pool1d(x, 6), x)
    assert np.array_equal(npref.adaptive_avg_pool1d(x[:, :5], 2), [[1.0, 3.0], [7.0, 9.0]])     # [0,3) and [2,5)
    assert np.array_equal(npref.adaptive_avg_pool1d(x[:, :2], 4), [[0, 0, 1, 1], [6, 6, 7, 7]])  # longer than the input
    assert npref.adaptive_avg_pool1d(x, 0).shape == (2, 0)


@pytest.mark.gpu
def test_device_adaptive_avg_pool1d_and_gather_bounds(ctx):
    from lele_amd import kernels as Kk
    from lele_amd._lib import LeleError
    rng = np.random.default_rng(8)
    for shape, o in (((2, 6), 3), ((3, 5, 171), 7), ((4, 1000), 999), ((2, 3), 8), ((1, 64, 93), 1), ((5, 17), 17)):
        x = rng.standard_normal(shape).astype(np.float32)
        got = Kk.adaptive_avg_pool1d(x, o, ctx=ctx)
        assert got.shape == shape[:-1] + (o,) and np.array_equal(got.numpy(), npref.adaptive_avg_pool1d(x, o)), (shape, o)
    # gather / gather_elements: an out-of-range index is the reference's slice-index panic (manipulation.rs:626-633) --
    # caught on the host for host-visible indices, reported at the next sync for device-resident ones (read clamped)
    emb = rng.standard_normal((16, 8)).astype(np.float32)
    with pytest.raises(LeleError, match="out of range"):
        Kk.gather(emb, np.array([3, 16], np.int64), 0, ctx=ctx)
    with pytest.raises(LeleError, match="out of range"):
        Kk.gather_elements(emb, np.full((16, 8), -17.0, np.float32), 0, ctx=ctx)
    bad = ctx.buf().upload(np.array([1, 99, -40], np.int64))
    got = Kk.gather(emb, bad, 0, ctx=ctx)
    with pytest.raises(LeleError, match="gather index out of range"):
        ctx.sync()
    ctx.sync()  # the flag is sticky until reported, then cleared
    assert np.array_equal(got.numpy()[0], emb[1])  # in-range rows are right, the others were clamped into the table
    ok = ctx.buf().upload(np.array([1, 15, -16], np.int64))
    assert np.array_equal(Kk.gather(emb, ok, 0, ctx=ctx).numpy(), emb[[1, 15, 0]])
    ctx.sync()
