"""Activations, element-wise broadcast ops, reductions, LayerNorm / Softmax / RMSNorm / BatchNorm.

CPU: the oracle (same AVX2 intrinsic sequences as lele's x86 kernels) against the reference's KATs and float64 math.
GPU: device vs oracle -- BIT-EXACT wherever the reference's x86 path is explicit arithmetic (polynomial SIMD bodies,
normalisation statistics, broadcast ops, reductions); <= 1e-4 relative where the reference calls libm in a scalar
tail (the device calls its own libm there)."""
import json
import math
import os

import numpy as np
import pytest

K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))
E = K["eltwise_regression"]


def _rng(lo, hi, step):
    return (np.arange(int(round(lo / step)), int(round(hi / step)) + 1, dtype=np.float32) * np.float32(step)).astype(
        np.float32)


def _f64(name, x):
    x = x.astype(np.float64)
    if name == "exp":
        return np.exp(x)
    if name == "sigmoid":
        return 1 / (1 + np.exp(-x))
    if name == "tanh":
        return np.tanh(x)
    if name == "silu":
        return x / (1 + np.exp(-x))
    if name == "erf":
        return np.array([math.erf(v) for v in x.ravel()]).reshape(x.shape)
    if name == "gelu":
        return x * 0.5 * (1 + np.array([math.erf(v / math.sqrt(2)) for v in x.ravel()]).reshape(x.shape))
    if name == "fast_gelu":
        return 0.5 * x * (1 + np.tanh(0.7978845608028654 * (x + 0.044715 * x ** 3)))
    if name == "relu":
        return np.maximum(x, 0)
    if name == "sqrt":
        return np.sqrt(x)


def test_oracle_kats(orc):
    k = K["softmax_ort"]
    assert np.abs(orc.softmax(np.array(k["x"], np.float32).reshape(k["x_shape"])).ravel() - k["expected"]).max() < k["tol"]
    k = K["softmax_simple"]
    assert np.abs(orc.softmax(np.array(k["x"], np.float32).reshape(k["x_shape"])).ravel() - k["expected"]).max() < k["tol"]
    k = K["layernorm_simple"]
    r = orc.layer_norm(np.array(k["x"], np.float32).reshape(k["x_shape"]), np.ones(3), np.zeros(3), -1, k["eps"])
    assert np.abs(r.ravel() - k["expected"]).max() < k["tol"]
    k = K["layernorm_stats"]
    r = orc.layer_norm(np.array(k["x"], np.float32).reshape(k["x_shape"]), np.ones(3), np.zeros(3), -1, k["eps"])
    var = 2.0 / 3.0
    assert abs(r[0].mean()) < 1e-5 and abs(r[0].std() - math.sqrt(var / (var + 1e-5))) < 1e-5
    assert np.array_equal(orc.unary("relu", np.array(K["relu"]["x"], np.float32)), np.array(K["relu"]["expected"], np.float32))
    x = _rng(-4, 4, 0.5)
    assert x.size == 17 and np.abs(orc.unary("silu", x) - _f64("silu", x)).max() < K["silu_17"]["tol"]
    x = _rng(-4.5, 4.5, 0.5)
    assert x.size == 19 and np.abs(orc.unary("erf", x) - _f64("erf", x)).max() < K["erf_19"]["tol"]
    for name, key, tol in (("sqrt", "sqrt_range", 1e-5), ("exp", "exp_range", 1e-4), ("tanh", "tanh_range", 1e-5),
                           ("sigmoid", "sigmoid_range", 1e-5), ("gelu", "gelu_range", 1e-5)):
        lo, hi, st = E[key]
        x = _rng(lo * st if key in ("sqrt_range", "exp_range", "sigmoid_range", "gelu_range") else lo,
                 hi * st if key in ("sqrt_range", "exp_range", "sigmoid_range", "gelu_range") else hi, st)
        ref = _f64(name, x)
        assert np.all(np.abs(orc.unary(name, x) - ref) <= tol * np.maximum(1.0, np.abs(ref))), name


def test_numpy_oracle_kats():
    from oracle import npref
    for op in ("sub", "div"):
        assert np.array_equal(npref.binary(op, np.array(E[op]["a"], np.float32), np.array(E[op]["b"], np.float32)),
                              np.array(E[op]["expected"], np.float32))
    c = E["clip"]
    assert np.array_equal(npref.clip(c["x"], c["min"], c["max"]), np.array(c["expected"], np.float32))
    for op in ("sum", "mean", "max", "l2"):
        r = E["reduce_" + op]
        got = npref.reduce(op, np.array(r["x"], np.float32).reshape(r["shape"]), r["axes"], False)
        assert np.allclose(got, r["expected"], atol=1e-6)
    r = E["reduce_sum"]
    assert npref.reduce("sum", np.array(r["x"], np.float32).reshape(r["shape"]), r["axes"], True).shape == (2, 1)
    assert np.array_equal(npref.unary_exact("neg", E["neg"]["x"]), np.array(E["neg"]["expected"], np.float32))


def test_oracle_norms_against_float64(orc):
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((7, 515)) * 2 + 0.5).astype(np.float32)
    g, b = rng.standard_normal(515).astype(np.float32), rng.standard_normal(515).astype(np.float32)
    x64 = x.astype(np.float64)
    ref = (x64 - x64.mean(1, keepdims=True)) / np.sqrt(x64.var(1, keepdims=True) + 1e-5) * g + b
    assert np.allclose(orc.layer_norm(x, g, b), ref, rtol=1e-4, atol=1e-5)
    e = np.exp(x64 - x64.max(1, keepdims=True))
    assert np.allclose(orc.softmax(x), e / e.sum(1, keepdims=True), rtol=1e-5, atol=1e-8)
    assert np.allclose(orc.rms_norm(x, g), x64 / np.sqrt((x64 ** 2).mean(1, keepdims=True) + 1e-5) * g, rtol=1e-5,
                       atol=1e-6)
    xb = rng.standard_normal((2, 5, 3, 11)).astype(np.float32)
    s, bb, m = (rng.standard_normal(5).astype(np.float32) for _ in range(3))
    v = rng.uniform(0.5, 2, 5).astype(np.float32)
    refb = (xb - m[None, :, None, None]) / np.sqrt(v + 1e-5)[None, :, None, None] * s[None, :, None, None] + bb[None, :, None, None]
    assert np.allclose(orc.batch_norm(xb, s, bb, m, v), refb, rtol=1e-5, atol=1e-5)


# ------------------------------------------------------------------------------------------------- GPU
pytest_gpu = pytest.mark.gpu
NAMES = {"exp": "exp", "sigmoid": "sigmoid", "tanh": "tanh_kernel", "silu": "silu", "erf": "erf", "gelu": "gelu",
         "fast_gelu": "fast_gelu", "relu": "relu", "sqrt": "sqrt"}


@pytest_gpu
@pytest.mark.parametrize("name", sorted(NAMES))
def test_device_unary_simd_body_bit_exact_tail_close(ctx, orc, name):
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(3)
    for n in (8, 64, 1000, 1003, 5, 21):
        x = (rng.standard_normal(n) * 4).astype(np.float32)
        if name == "sqrt":
            x = np.abs(x)
        x[: min(n, 4)] = np.array([0.0, -0.0, 100.0, -100.0], np.float32)[: min(n, 4)] if name != "sqrt" else 0.0
        ref = orc.unary(name, x)
        got = getattr(Kk, NAMES[name])(x, ctx=ctx).numpy()
        body = n & ~7
        # equal_nan: the reference's polynomial tanh gives NaN for x <= -44.4 ((1-inf)/(1+inf)); reproduced as is
        assert np.array_equal(got[:body], ref[:body], equal_nan=True), (name, n)
        gt, rt = got[body:], ref[body:]
        fin = np.isfinite(rt)
        assert np.array_equal(gt[~fin], rt[~fin])
        assert np.all(np.abs(gt[fin] - rt[fin]) <= 1e-4 * np.abs(rt[fin]) + 1e-7), (name, n, gt, rt)


@pytest_gpu
def test_device_unary_libm_ops(ctx):
    from lele_amd import kernels as Kk
    from oracle import npref
    rng = np.random.default_rng(4)
    x = (rng.standard_normal((3, 37)) * 3).astype(np.float32)
    for name in ("neg", "reciprocal", "not_", "abs", "floor", "ceil"):
        xx = np.where(rng.uniform(size=x.shape) < 0.2, 0, x).astype(np.float32) if name == "not_" else x
        assert np.array_equal(getattr(Kk, name)(xx, ctx=ctx).numpy(), npref.unary_exact(name, xx)), name
    for name in ("log", "sin", "cos", "softplus"):
        xx = np.abs(x) + 0.1 if name == "log" else x
        ref = npref.unary_exact(name, xx)
        assert np.all(np.abs(getattr(Kk, name)(xx, ctx=ctx).numpy() - ref) <= 1e-4 * np.abs(ref) + 1e-6), name


@pytest_gpu
def test_device_binary_broadcast_bit_exact(ctx):
    from lele_amd import kernels as Kk
    from oracle import npref
    rng = np.random.default_rng(5)
    shapes = [((4,), (4,)), ((2, 3, 4), (2, 3, 4)), ((2, 3, 4), (1,)), ((1,), (5, 2)), ((2, 3, 4), (4,)),
              ((1, 16, 8, 8), (16, 1, 1)), ((3, 1, 5), (1, 4, 1)), ((6, 1), (1, 7)), ((2, 1, 4), (3, 1))]
    for sa, sb in shapes:
        a = (rng.standard_normal(sa) * 3).astype(np.float32)
        b = (rng.standard_normal(sb) * 3).astype(np.float32)
        for name in ("add", "sub", "mul", "div", "max", "min", "equal", "less", "greater", "prelu", "mod_f32", "and_",
                     "or_"):
            bb = np.where(np.abs(b) < 0.3, 0, b).astype(np.float32) if name in ("mod_f32", "and_", "or_", "equal") else b
            got = getattr(Kk, name)(a, bb, ctx=ctx)
            ref = npref.binary(name, a, bb)
            assert got.shape == ref.shape and np.array_equal(got.numpy(), ref, equal_nan=True), (name, sa, sb)
    a = np.abs(rng.standard_normal((5, 6))).astype(np.float32) + 0.1
    b = rng.uniform(-2, 3, (6,)).astype(np.float32)
    ref = npref.binary("pow", a, b)
    assert np.all(np.abs(Kk.pow(a, b, ctx=ctx).numpy() - ref) <= 1e-4 * np.abs(ref))
    ai = rng.integers(-50, 50, (3, 4)).astype(np.int64)
    bi = rng.integers(1, 9, (4,)).astype(np.int64)
    for name in ("add", "sub", "mul", "div"):
        got = getattr(Kk, name)(ai, bi, ctx=ctx)
        assert got.dtype == np.int64 and np.array_equal(got.numpy(), npref.binary(name, ai, bi)), name
    for op in ("sub", "div"):
        assert np.array_equal(getattr(Kk, op)(np.array(E[op]["a"], np.float32), np.array(E[op]["b"], np.float32),
                                              ctx=ctx).numpy(), np.array(E[op]["expected"], np.float32))


@pytest_gpu
def test_device_where_clip_reduce_bit_exact(ctx):
    import lele_amd
    from lele_amd import kernels as Kk
    from oracle import npref
    rng = np.random.default_rng(6)
    cond = (rng.uniform(size=(2, 1, 4)) > 0.5).astype(np.float32)
    x, y = rng.standard_normal((2, 3, 4)).astype(np.float32), rng.standard_normal((4,)).astype(np.float32)
    assert np.array_equal(Kk.where_op(cond, x, y, ctx=ctx).numpy(), npref.where_op(cond, x, y))
    f32 = lambda v: np.array(v, np.float32)
    assert np.array_equal(Kk.where_op(f32([1, 0, 0, 1]), f32([1, 2, 3, 4]), f32([5, 6, 7, 8]), ctx=ctx).numpy().ravel(),
                          np.array([1, 6, 7, 4], np.float32))  # kernel_accuracy.rs:173-191
    with pytest.raises(lele_amd.LeleError, match="f32"):
        Kk.where_op(np.array([1, 0], np.int64), f32([1, 2]), f32([3, 4]), ctx=ctx)
    c = E["clip"]
    assert np.array_equal(Kk.clip(np.array(c["x"], np.float32), [c["min"]], [c["max"]], ctx=ctx).numpy(),
                          np.array(c["expected"], np.float32))
    assert np.array_equal(Kk.clip(x, None, [0.5], ctx=ctx).numpy(), npref.clip(x, None, 0.5))
    big = (rng.standard_normal((3, 50, 7)) * 10).astype(np.float32)
    for op, fn in (("sum", Kk.reduce_sum), ("mean", Kk.reduce_mean), ("max", Kk.reduce_max), ("l2", Kk.reduce_l2)):
        for axes in ([1], [0, 2], [-1], [0, 1, 2]):
            for kd in (False, True):
                got = fn(big, axes, kd, ctx=ctx)
                ref = npref.reduce(op, big, axes, kd)
                assert got.shape == ref.shape and np.array_equal(got.numpy(), ref), (op, axes, kd)
    # max / min over a long contiguous last axis take the 16-lanes-per-row kernel: same value as the sequential scan, signed zeros and
    # NaN included (the first of equal values wins; NaN never does)
    wide = (rng.standard_normal((5, 37, 80)) * 3).astype(np.float32)
    wide[0, 0, :] = -0.0
    wide[0, 0, 17] = 0.0
    wide[0, 1, :] = 0.0
    wide[0, 1, 40] = -0.0
    wide[1, 2, 5] = np.nan
    wide[1, 3, :] = np.nan
    for axes, kd in (([-1], False), ([2], True)):
        got, ref = Kk.reduce_max(wide, axes, kd, ctx=ctx).numpy(), npref.reduce("max", wide, axes, kd)
        assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    # few long rows (a global max is ONE row): pieces of a row per workgroup, then a merge -- the first of equal values still wins
    longrow = (rng.standard_normal((3, 300001)) * 3).astype(np.float32)
    longrow[0, :] = np.minimum(longrow[0, :], 0.0)
    longrow[0, 123456] = -0.0
    longrow[0, 250000] = 0.0                      # max = +/-0: the first zero in scan order decides the sign
    zeros = np.flatnonzero(longrow[0] == 0.0)
    longrow[1, 77] = np.nan
    longrow[2, :] = np.nan
    for axes, kd in (([-1], False), ([1], True)):
        got, ref = Kk.reduce_max(longrow, axes, kd, ctx=ctx).numpy(), npref.reduce("max", longrow, axes, kd)
        assert got.shape == ref.shape and np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert np.signbit(Kk.reduce_max(longrow, [-1], False, ctx=ctx).numpy()[0]) == np.signbit(longrow[0, zeros[0]])
    flat = Kk.reduce_max(longrow[:2].reshape(-1).copy(), [0], False, ctx=ctx).numpy()
    assert np.array_equal(flat.view(np.uint32), npref.reduce("max", longrow[:2].reshape(-1), [0], False).view(np.uint32))
    with pytest.raises(lele_amd.LeleError, match="broadcastable"):
        Kk.add(np.zeros((2, 3), np.float32), np.zeros((4,), np.float32), ctx=ctx)


@pytest_gpu
@pytest.mark.parametrize("shape", [(1, 3), (2, 4), (1, 504, 512), (3, 7, 515), (5, 33), (2, 9, 8), (4, 1), (1, 4, 504, 504),
                                   (1, 2, 400, 400), (6, 1000)])
def test_device_norms_bit_exact(ctx, orc, shape):
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(sum(shape))
    x = (rng.standard_normal(shape) * 2 + 0.3).astype(np.float32)
    n = shape[-1]
    g = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    b = (0.1 * rng.standard_normal(n)).astype(np.float32)
    assert np.array_equal(Kk.layer_norm(x, g, b, -1, 1e-5, ctx=ctx).numpy(), orc.layer_norm(x, g, b, -1, 1e-5))
    assert np.array_equal(Kk.rms_norm(x, g, -1, 1e-6, ctx=ctx).numpy(), orc.rms_norm(x, g, -1, 1e-6))
    got, ref = Kk.softmax(x, -1, ctx=ctx).numpy(), orc.softmax(x)
    if n % 8 == 0:
        assert np.array_equal(got, ref)
    else:
        assert np.all(np.abs(got - ref) <= 1e-4 * np.abs(ref) + 1e-9)  # libm exp in the scalar tail
    assert np.all(np.abs(got.sum(-1) - 1) < 1e-5)


@pytest_gpu
def test_device_norm_kats_axis_and_errors(ctx, orc):
    import lele_amd
    from lele_amd import kernels as Kk
    k = K["softmax_ort"]
    assert np.abs(Kk.softmax(np.array(k["x"], np.float32).reshape(k["x_shape"]), -1, ctx=ctx).data - k["expected"]).max() < k["tol"]
    k = K["layernorm_simple"]
    r = Kk.layer_norm(np.array(k["x"], np.float32).reshape(k["x_shape"]), np.ones(3, np.float32), np.zeros(3, np.float32),
                      -1, k["eps"], ctx=ctx)
    assert np.abs(r.data - k["expected"]).max() < k["tol"]
    rng = np.random.default_rng(8)
    x = rng.standard_normal((2, 3, 4, 5)).astype(np.float32)  # normalise over the trailing [4,5]
    g, b = rng.standard_normal((4, 5)).astype(np.float32), rng.standard_normal((4, 5)).astype(np.float32)
    assert np.array_equal(Kk.layer_norm(x, g, b, 2, 1e-5, ctx=ctx).numpy(), orc.layer_norm(x, g, b, 2, 1e-5))
    with pytest.raises(lele_amd.LeleError, match="last dimension"):
        Kk.softmax(x, 1, ctx=ctx)  # norm.rs:218 unimplemented!
    s, bb, m = (rng.standard_normal(3).astype(np.float32) for _ in range(3))
    v = rng.uniform(0.5, 2, 3).astype(np.float32)
    assert np.array_equal(Kk.batch_norm(x, s, bb, m, v, 1e-5, ctx=ctx).numpy(), orc.batch_norm(x, s, bb, m, v))
    x2 = rng.standard_normal((6, 3)).astype(np.float32)
    assert np.array_equal(Kk.batch_norm(x2, s, bb, m, v, 1e-5, ctx=ctx).numpy(), orc.batch_norm(x2, s, bb, m, v))


@pytest.mark.gpu
def test_i64_result_comparisons_and_min_max(ctx):
    # math.rs:56, 1201-1235, 2161: the i64-valued comparison variants lele's emitters use for shape arithmetic
    from lele_amd import kernels as K
    a = np.array([[1.0, 2.5, -3.0], [4.0, 2.5, 7.9]], np.float32)
    b = np.array([1.0, 2.0, -3.0], np.float32)
    r = K.equal_i64(a, b, ctx=ctx).numpy()
    assert r.dtype == np.int64 and np.array_equal(r, (a == b).astype(np.int64))
    ai, bi = np.array([[1, 2, 3], [3, 2, 1]], np.int64), np.array([1, 2, 1], np.int64)
    assert np.array_equal(K.equal_i64(ai, bi, ctx=ctx).numpy(), (ai == bi).astype(np.int64))
    assert np.array_equal(K.less_i64(ai, bi, ctx=ctx).numpy(), (ai < bi).astype(np.int64))
    assert np.array_equal(K.less_i64(a, b, ctx=ctx).numpy(), (a < b).astype(np.int64))
    # `v as i64` truncates toward zero before comparing: 2.5 and 2.0 both become 2
    assert np.array_equal(K.equal_i64_f32_r(a, b, ctx=ctx).numpy(), (a.astype(np.int64) == b.astype(np.int64)).astype(np.int64))
    assert np.array_equal(K.equal_i64_f32_r_i64(ai, b, ctx=ctx).numpy(), (ai == b.astype(np.int64)).astype(np.int64))
    assert np.array_equal(K.equal_i64_f32_lhs(a, bi, ctx=ctx).numpy(), (a.astype(np.int64) == bi).astype(np.int64))
    assert K.min_max(a, ctx=ctx) == (-3.0, float(np.float32(7.9)))
