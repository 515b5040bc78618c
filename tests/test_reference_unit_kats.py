"""The reference's in-source unit tests for the data-movement / pooling / resize / conv_transpose / mod / LSTM operators,
transcribed as data (inputs and expected outputs), run against the CPU oracle (always) and the HIP operators (-m gpu).
Sources: src/kernels/math.rs:2477-2512 (mod_f32), src/kernels/conv2d.rs:3391-3749 (conv_transpose, resize_nearest,
max_pool2d), tests/regression_kernels.rs:977-997 (LSTM shapes), src/kernels/activations.rs:17-26."""
import numpy as np
import pytest

from oracle import npref
from oracle import pyoracle as O

MOD_KATS = [  # (a, a_shape, b, b_shape, expected)  math.rs:2477-2512; x % 0 -> 0
    ([9, 13, 7, 0, 25], [5], [5, 5, 3, 2, 10], [5], [4, 3, 1, 0, 5]),
    ([3, 9, 13, 18, 23], [5], [5], [], [3, 4, 3, 3, 3]),
    ([3, 9, 13, 18], [2, 2], [5], [], [3, 4, 3, 3]),
    ([5, 10], [2], [0], [], [0, 0]),
]
CT_SHAPES = [  # (x shape, w shape, pads, strides, expected output shape)  conv2d.rs:3391-3473
    ((1, 64, 80, 80), (64, 64, 2, 2), [0, 0, 0, 0], [2, 2], (1, 64, 160, 160)),
    ((1, 32, 40, 40), (32, 32, 3, 3), [1, 1, 1, 1], [2, 2], (1, 32, 79, 79)),
    ((1, 16, 10, 10), (16, 16, 3, 3), [0, 0, 0, 0], [1, 1], (1, 16, 12, 12)),
]


def _ops(device, ctx):
    """the same call surface for the oracle and the device mirror"""
    if not device:
        class Orc:
            mod_f32 = staticmethod(lambda a, b: npref.binary("mod_f32", a, b))
            conv_transpose = staticmethod(lambda x, w, b, d, g, p, s: O.conv_transpose(x, w, b, d, g, p, s))
            max_pool2d = staticmethod(lambda x, k, s, p, d, c: npref.max_pool2d(x, k, s, p, d, c))
            lstm = staticmethod(lambda x, w, r, b: O.lstm(x, w, r, b))

            @staticmethod
            def resize_nearest(x, scales, sizes, mode):
                if sizes is not None:
                    if sizes[2] <= 0 or sizes[3] <= 0:
                        raise ValueError("sizes H and W must be positive")
                    oh, ow = sizes[2], sizes[3]
                else:
                    oh, ow = int(x.shape[2] * scales[2]), int(x.shape[3] * scales[3])
                return npref.resize_nearest(x, oh, ow, mode == "asymmetric")
        return Orc
    from lele_amd import kernels as K

    class Dev:
        mod_f32 = staticmethod(lambda a, b: K.mod_f32(a, b, ctx=ctx).numpy())
        conv_transpose = staticmethod(lambda *a: K.conv_transpose(*a, ctx=ctx).numpy())
        max_pool2d = staticmethod(lambda *a: K.max_pool2d(*a, ctx=ctx).numpy())
        resize_nearest = staticmethod(lambda x, sc, sz, mode: K.resize_nearest(x, sc, sz, mode, ctx=ctx).numpy())
        lstm = staticmethod(lambda x, w, r, b: tuple(t.numpy() for t in K.lstm(x, w, r, b, ctx=ctx)))
    return Dev


def _run(ops):
    f = np.float32
    for a, ash, b, bsh, want in MOD_KATS:
        got = ops.mod_f32(np.array(a, f).reshape(ash), np.array(b, f).reshape(bsh))
        assert got.shape == tuple(ash) and np.array_equal(got.ravel(), np.array(want, f))
    for xs, ws, pads, strides, oshape in CT_SHAPES:  # zeros in, ones weights, zero bias -> zeros of the stated shape
        got = ops.conv_transpose(np.zeros(xs, f), np.ones(ws, f), np.zeros(ws[1], f), [1, 1], 1, pads, strides)
        assert got.shape == oshape and not got.any()
    got = ops.conv_transpose(np.array([1.0], f).reshape(1, 1, 1, 1), np.array([2.0], f).reshape(1, 1, 1, 1), None, [1], 1,
                             [0, 0, 0, 0], [1, 1])
    assert got.shape == (1, 1, 1, 1) and abs(float(got.ravel()[0]) - 2.0) < 1e-6           # conv2d.rs:3476-3495
    x22 = np.array([1, 2, 3, 4], f).reshape(1, 1, 2, 2)
    assert np.array_equal(ops.resize_nearest(x22, [1, 1, 1, 1], None, "asymmetric"), x22)   # :3500
    up = ops.resize_nearest(x22, [1.0, 1.0, 2.0, 2.0], None, "asymmetric")                  # :3520
    assert np.array_equal(up.ravel(), np.array([1, 1, 2, 2, 1, 1, 2, 2, 3, 3, 4, 4, 3, 3, 4, 4], f))
    assert ops.resize_nearest(x22, None, [1, 1, 3, 3], "asymmetric").shape == (1, 1, 3, 3)  # :3551
    assert ops.resize_nearest(np.arange(8, dtype=f).reshape(1, 2, 2, 2), [1.0, 1.0, 2.0, 2.0], None, "asymmetric").shape == (1, 2, 4, 4)
    hp = ops.resize_nearest(x22, [1.0, 1.0, 2.0, 2.0], None, "half_pixel")                  # :3579
    assert hp.shape == (1, 1, 4, 4) and hp.min() >= 1.0 and hp.max() <= 4.0
    big = ops.resize_nearest(np.array([42.0], f).reshape(1, 1, 1, 1), None, [1, 1, 100, 100], "asymmetric")  # :3600
    assert big.shape == (1, 1, 100, 100) and (big == 42.0).all()
    with pytest.raises(Exception, match="must be positive"):                                # :3619-3625
        ops.resize_nearest(np.array([1.0], f).reshape(1, 1, 1, 1), None, [1, 1, -1, 10], "asymmetric")
    x16 = np.arange(16, dtype=f).reshape(1, 1, 4, 4)
    assert np.array_equal(ops.max_pool2d(x16, [2, 2], [2, 2], [0, 0, 0, 0], [1, 1], False).ravel(), np.array([5, 7, 13, 15], f))
    assert np.array_equal(ops.max_pool2d(np.arange(9, dtype=f).reshape(1, 1, 3, 3), [2, 2], [1, 1], [0, 0, 0, 0], [1, 1], False).ravel(),
                          np.array([4, 5, 7, 8], f))
    pp = ops.max_pool2d(x22, [2, 2], [1, 1], [1, 1, 1, 1], [1, 1], False)                   # :3680
    assert pp.shape == (1, 1, 3, 3) and pp.ravel()[0] == 1.0 and pp.ravel()[4] == 4.0 and pp.ravel()[8] == 4.0
    assert ops.max_pool2d(np.arange(32, dtype=f).reshape(1, 2, 4, 4), [2, 2], [2, 2], [0, 0, 0, 0], [1, 1], False).shape == (1, 2, 2, 2)
    ones = ops.max_pool2d(np.ones((1, 32, 80, 80), f), [2, 2], [2, 2], [0, 0, 0, 0], [1, 1], False)  # :3724
    assert ones.shape == (1, 32, 40, 40) and (ones == 1.0).all()
    # tests/regression_kernels.rs:977-997: LSTM single step, shapes + finiteness
    hs, isz = 4, 3
    w = (np.arange(4 * hs * isz, dtype=f) * f(0.01)).reshape(1, 4 * hs, isz)
    r = (np.arange(4 * hs * hs, dtype=f) * f(0.02) - f(0.1)).reshape(1, 4 * hs, hs)
    b = (np.arange(8 * hs, dtype=f) * f(0.005)).reshape(1, 8 * hs)
    y, h, c = ops.lstm(np.array([0.1, -0.2, 0.3], f).reshape(1, 1, isz), w, r, b)
    assert y.shape == (1, 1, 1, hs) and h.shape == (1, 1, hs) and c.shape == (1, 1, hs)
    assert np.isfinite(y).all() and np.isfinite(c).all()
    return y, h, c


def test_oracle_reproduces_the_reference_unit_tests():
    _run(_ops(False, None))
    # activations.rs:17-26: sigmoid(0) = 0.5 (libm tail path of the unary kernel: one element)
    assert abs(float(O.unary("sigmoid", np.zeros(1, np.float32))[0]) - 0.5) < 1e-6


@pytest.mark.gpu
def test_device_reproduces_the_reference_unit_tests(ctx):
    yd = _run(_ops(True, ctx))
    yo = _run(_ops(False, None))
    for a, b in zip(yd, yo):
        assert np.abs(a - b).max() <= 1e-6
