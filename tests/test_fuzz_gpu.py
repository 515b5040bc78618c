"""Randomised shape sweep of the kernels with the most code paths (tile selection, small-problem split-K, tap-major and
phase-decomposed convolutions, LDS-tiled / vectorised / generic copies, quantised GEMM variants) against the oracle.
Seeded: every run checks the same cases (LELE_FUZZ_SEED=<n> shifts every seed for an extra sweep)."""
import os

import numpy as np
import pytest

from oracle import npref
from oracle import pyoracle as O

pytestmark = pytest.mark.gpu
RTOL = 1e-4
SEED = int(os.environ.get("LELE_FUZZ_SEED", "0"))


def _close(got, want, what):
    from parity import close_f32
    close_f32(got, want, RTOL, what)


def test_fuzz_matmul_gemm(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(101 + SEED)
    for it in range(60):
        m, k, n = (int(rng.integers(1, 300)) for _ in range(3))
        if it % 7 == 0:
            m, n = int(rng.integers(300, 700)), int(rng.integers(300, 700))   # tiled kernels
        ba = int(rng.choice([1, 1, 2, 5]))
        bb = int(rng.choice([1, ba]))
        a = rng.standard_normal((ba, m, k)).astype(np.float32) if ba > 1 or it % 2 else rng.standard_normal((m, k)).astype(np.float32)
        b = rng.standard_normal((bb, k, n)).astype(np.float32) if bb > 1 else rng.standard_normal((k, n)).astype(np.float32)
        _close(K.matmul(a, b, ctx=ctx).numpy(), O.matmul(a, b), "matmul %s x %s" % (a.shape, b.shape))
    for it in range(30):
        m, k, n = (int(rng.integers(1, 200)) for _ in range(3))
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        a = rng.standard_normal((k, m) if ta else (m, k)).astype(np.float32)
        b = rng.standard_normal((n, k) if tb else (k, n)).astype(np.float32)
        cshape = [(m, n), (n,), (m, 1), (1,), None][it % 5]
        c = None if cshape is None else rng.standard_normal(cshape).astype(np.float32)
        alpha, beta = float(rng.uniform(-2, 2)), float(rng.uniform(-2, 2))
        _close(K.gemm(a, b, c, alpha, beta, ta, tb, ctx=ctx).numpy(), O.gemm(a, b, c, alpha, beta, ta, tb),
               "gemm ta=%s tb=%s c=%s" % (ta, tb, cshape))


def test_fuzz_conv2d_conv_transpose(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(202 + SEED)
    for it in range(70):
        g = int(rng.choice([1, 1, 1, 2, 4]))
        icg, ocg = int(rng.choice([1, 3, 4, 8, 12, 16])), int(rng.choice([1, 2, 8, 16, 33]))
        c, oc = g * icg, g * ocg
        if it % 9 == 0:
            g = c = oc = int(rng.choice([8, 24]))  # depthwise
            icg = 1
        n, h, w_ = int(rng.integers(1, 4)), int(rng.integers(5, 30)), int(rng.integers(5, 30))
        kh, kw = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        sh, sw, dh, dw = (int(rng.integers(1, 3)) for _ in range(4))
        pads = [int(v) for v in rng.integers(0, 3, 4)]
        if h + pads[0] + pads[2] < dh * (kh - 1) + 1 or w_ + pads[1] + pads[3] < dw * (kw - 1) + 1:
            continue
        x = rng.standard_normal((n, c, h, w_)).astype(np.float32)
        wt = (rng.standard_normal((oc, c // g, kh, kw)) * 0.3).astype(np.float32)
        b = rng.standard_normal(oc).astype(np.float32) if it % 3 else None
        act = [None, "relu", "silu"][it % 3]
        fn = {"silu": K.conv2d_silu, "relu": lambda *a, **k: K.conv2d_fused(*a, relu=True, **k), None: K.conv2d}[act]
        _close(fn(x, wt, b, [dh, dw], g, pads, [sh, sw], ctx=ctx).numpy(), O.conv2d(x, wt, b, [dh, dw], g, pads, [sh, sw], act),
               "conv2d it=%d %s w%s g=%d pads=%s s=%s d=%s act=%s" % (it, x.shape, wt.shape, g, pads, (sh, sw), (dh, dw), act))
    for it in range(40):
        c, oc = int(rng.choice([1, 3, 4, 8, 13])), int(rng.choice([1, 2, 7, 16]))
        n, h, w_ = int(rng.integers(1, 3)), int(rng.integers(2, 12)), int(rng.integers(2, 12))
        kh, kw = int(rng.integers(1, 5)), int(rng.integers(1, 5))
        sh, sw, dh, dw = (int(rng.integers(1, 4)) for _ in range(4))
        pads = [int(v) for v in rng.integers(0, 2, 4)]
        if (h - 1) * sh - pads[0] - pads[2] + dh * (kh - 1) + 1 <= 0 or (w_ - 1) * sw - pads[1] - pads[3] + dw * (kw - 1) + 1 <= 0:
            continue
        x = rng.standard_normal((n, c, h, w_)).astype(np.float32)
        wt = (rng.standard_normal((c, oc, kh, kw)) * 0.3).astype(np.float32)
        b = rng.standard_normal(oc).astype(np.float32) if it % 2 else None
        _close(K.conv_transpose(x, wt, b, [dh, dw], 1, pads, [sh, sw], ctx=ctx).numpy(),
               O.conv_transpose(x, wt, b, [dh, dw], 1, pads, [sh, sw]),
               "conv_transpose it=%d %s w%s pads=%s s=%s d=%s" % (it, x.shape, wt.shape, pads, (sh, sw), (dh, dw)))


def test_fuzz_quantized_linear(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(303 + SEED)
    for it in range(40):
        batch = int(rng.choice([1, 1, 2, 3]))
        m, k, n = int(rng.integers(1, 200)), int(rng.integers(1, 300)), int(rng.integers(1, 200))
        if it % 8 == 0:
            m, n = int(rng.integers(600, 900)), int(rng.integers(500, 700))  # tiled i8 kernels
        x = (rng.standard_normal((batch, m, k)) * rng.uniform(0.1, 5)).astype(np.float32)
        w = rng.integers(0, 256, (k, n)).astype(np.float32)
        ws = (np.abs(rng.standard_normal(n if it % 2 else 1)) * 0.01 + 0.002).astype(np.float32)
        wz = np.array([float(rng.integers(100, 150))], np.float32)
        b = rng.standard_normal(n).astype(np.float32) if it % 3 else None
        relu = bool(it % 2)
        got = K.fused_quantized_linear(x, w, ws, wz, b, relu, ctx=ctx).numpy()
        want = O.fused_quantized_linear(x, w, ws, wz, b, relu)
        assert np.array_equal(got, want), "fused_quantized_linear it=%d %s x %s: not bit-exact" % (it, x.shape, w.shape)


def test_fuzz_strided_copies(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(404 + SEED)
    for it in range(80):
        rank = int(rng.integers(1, 6))
        shape = [int(rng.integers(1, 9)) for _ in range(rank)]
        if it % 5 == 0:
            shape[-1] = int(rng.choice([32, 64, 128]))   # vectorisable / tiled paths
            shape[0] = int(rng.choice([16, 40]))
        dt = np.float32 if it % 4 else np.int64
        x = (rng.standard_normal(shape) * 100).astype(dt)
        perm = [int(v) for v in rng.permutation(rank)]
        assert np.array_equal(K.transpose(x, perm, ctx=ctx).numpy(), np.transpose(x, perm)), ("transpose", shape, perm)
        starts = [int(rng.integers(-s - 1, s + 1)) for s in shape]
        ends = [int(rng.integers(-s - 1, s + 2)) for s in shape]
        steps = [int(rng.choice([1, 1, 2, -1, 3])) for _ in shape]
        axes = list(range(rank))
        assert np.array_equal(K.slice(x, starts, ends, axes, steps, ctx=ctx).numpy(), npref.slice_(x, starts, ends, axes, steps)), (
            "slice", shape, starts, ends, steps)
        reps = [int(rng.integers(1, 3)) for _ in shape]
        assert np.array_equal(K.tile(x, reps, ctx=ctx).numpy(), np.tile(x, reps)), ("tile", shape, reps)
        ax = int(rng.integers(0, rank))
        parts = [x, x[tuple(slice(0, max(1, s // 2)) if d == ax else slice(None) for d, s in enumerate(shape))]]
        assert np.array_equal(K.concat(parts, ax, ctx=ctx).numpy(), np.concatenate(parts, ax)), ("concat", shape, ax)


def test_fuzz_broadcast_reduce_norm_pad_gather(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(505 + SEED)
    for it in range(60):
        rank = int(rng.integers(1, 5))
        shape = [int(rng.integers(1, 8)) for _ in range(rank)]
        a = rng.standard_normal(shape).astype(np.float32)
        bshape = [s if rng.integers(0, 2) else 1 for s in shape][int(rng.integers(0, rank)):]  # numpy-style broadcast partner
        b = (rng.standard_normal(bshape) + 2.5).astype(np.float32)
        for name in ("add", "sub", "mul", "div", "max", "min", "less", "greater", "equal"):
            got = getattr(K, name)(a, b, ctx=ctx).numpy()
            assert np.array_equal(got, npref.binary(name, a, b)), (name, shape, bshape)
        axes = sorted(set(int(v) for v in rng.integers(0, rank, int(rng.integers(1, rank + 1)))))
        keep = bool(it % 2)
        for op, fn in (("sum", K.reduce_sum), ("mean", K.reduce_mean), ("max", K.reduce_max), ("l2", K.reduce_l2)):
            assert np.array_equal(fn(a, axes, keep, ctx=ctx).numpy(), npref.reduce(op, a, axes, keep)), (op, shape, axes, keep)
        pads = [int(v) for v in rng.integers(0, 3, 2 * rank)]
        mode = ["constant", "edge", "reflect"][it % 3]
        if mode == "reflect" and any(p >= s for p, s in zip(pads[:rank], shape)) or any(p >= s for p, s in zip(pads[rank:], shape)):
            mode = "constant"
        assert np.array_equal(K.pad(a, pads, np.array([1.5], np.float32), mode, ctx=ctx).numpy(), npref.pad(a, pads, 1.5, mode)), (
            "pad", shape, pads, mode)
        ax = int(rng.integers(0, rank))
        idx = rng.integers(0, shape[ax], (int(rng.integers(1, 5)),)).astype(np.float32)
        assert np.array_equal(K.gather(a, idx, ax, ctx=ctx).numpy(), npref.gather(a, idx, ax)), ("gather", shape, ax)
    for it in range(30):
        outer, n = int(rng.integers(1, 40)), int(rng.choice([1, 7, 8, 31, 32, 33, 80, 171, 255, 256, 257, 504, 512, 600, 1024, 1500]))
        x = (rng.standard_normal((outer, n)) * 3).astype(np.float32)
        g, bta = rng.standard_normal(n).astype(np.float32), rng.standard_normal(n).astype(np.float32)
        assert np.array_equal(K.layer_norm(x, g, bta, -1, 1e-5, ctx=ctx).numpy(), O.layer_norm(x, g, bta, -1, 1e-5)), ("ln", outer, n)
        got, want = K.softmax(x, -1, ctx=ctx).numpy(), O.softmax(x, -1)
        body = n & ~7
        assert np.array_equal(got[:, :body], want[:, :body]) and np.abs(got - want).max() <= 1e-6, ("softmax", outer, n)
        assert np.array_equal(K.rms_norm(x, g, -1, 1e-6, ctx=ctx).numpy(), O.rms_norm(x, g, -1, 1e-6)), ("rms", outer, n)
