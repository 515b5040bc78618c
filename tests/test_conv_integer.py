"""ConvInteger family (conv2d.rs:1507-2761).  On x86 lele centres both operands in f32 and runs
its f32 GEMM over an im2col whose PADDED cells hold -x_zp (a padded cell is the u8 value 0, conv2d.rs:2025): the oracle is
pyoracle.conv_integer (+ the numpy DynamicQuantizeLinear of npref).  The device multiplies the u8 / i8 codes on the i8 matrix cores
and applies the zero points algebraically: exact integers, so results are compared for EQUALITY wherever the reference's own f32
sums are exact (every partial sum below 2^24)."""
import numpy as np
import pytest

from oracle import npref
from oracle import pyoracle as O



def _close(got, want, tol=1e-4):
    """element by element: |got - want| <= tol * |want| + tol * rms(want) (the form of tests/test_conv_rnn.py::_close)"""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape
    floor = tol * float(np.sqrt(np.mean(np.square(want)))) + 1e-7
    bad = np.abs(got - want) > tol * np.abs(want) + floor
    assert not bad.any(), "%d of %d elements outside %g (max abs diff %.3e)" % (int(bad.sum()), want.size, tol, float(np.abs(got - want).max()))

def _u8(rng, shape):
    return rng.integers(0, 256, shape).astype(np.float32)


def test_oracle_conv_integer_is_exact_integer_arithmetic():
    rng = np.random.default_rng(0)
    x, w = _u8(rng, (1, 3, 6, 6)), _u8(rng, (4, 3, 3, 3))
    ref = O.conv_integer(x, w, 120.0, 128.0, [1, 1], 1, [1, 1, 1, 1], [1, 1])
    # small K: every partial sum is an integer below 2^24 -> the f32 result is the exact integer convolution.  Padded cells are the
    # u8 value 0, i.e. -120 once centred (im2col_with_zp, conv2d.rs:2025: "pad value is (0 - x_zp)")
    xi, wi = x.astype(np.int64) - 120, w.astype(np.int64) - 128
    xp = np.pad(xi, ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-120)
    want = np.zeros((1, 4, 6, 6), np.int64)
    for o in range(4):
        for y in range(6):
            for xx in range(6):
                want[0, o, y, xx] = (xp[0, :, y:y + 3, xx:xx + 3] * wi[o]).sum()
    assert np.array_equal(ref, want.astype(np.float32))
    s, z = npref.dql_params([np.array([-1.0, 0.5, 3.0], np.float32)])
    assert s == np.float32(4.0) / np.float32(255.0) and z == np.float32(64.0)  # round(1 / (4/255)) = round(63.75)
    assert npref.dql_quantize(np.array([-1.0, 0.0, 3.0], np.float32), s, z).tolist() == [0.0, 64.0, 255.0]


def _exact(x, xz, w, wz):
    """can the reference's f32 sums round?  Not while K * max|x - zx| * max|w - zw| stays below 2^24: then every partial sum of
    every summation order is an exactly representable integer and the device's i32 arithmetic must EQUAL the oracle"""
    k = w.shape[1] * w.shape[2] * w.shape[3]
    return k * float(np.abs(x - xz).max()) * float(np.abs(w - wz).max()) < 2.0 ** 24


@pytest.mark.gpu
def test_device_conv_integer_family(ctx):
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(1)
    exact = 0
    cases = ((2, 8, 12, 16, 3, 1, 1, 1), (1, 16, 9, 8, 1, 1, 1, 0), (1, 6, 11, 9, 3, 2, 3, 1), (1, 64, 20, 64, 3, 1, 1, 1), (3, 40, 17, 70, 3, 2, 1, 1),
             (1, 33, 14, 5, 5, 1, 1, 2), (2, 96, 10, 40, 1, 1, 1, 0))
    for (n, c, h, oc, k, st, g, p) in cases:
        x, w = _u8(rng, (n, c, h, h + 3)), _u8(rng, (oc, c // g, k, k))
        for xz, wz in ((128.0, 128.0), (0.0, 113.0), (7.0, 0.0), (None, None), (255.0, 255.0)):
            zx = None if xz is None else np.array([xz], np.float32)
            zw = None if wz is None else np.array([wz], np.float32)
            got = K.conv_integer(x, Weight(w), zx, zw, [1, 1], g, [p, p, p, p], [st, st], ctx=ctx).numpy()
            want = O.conv_integer(x, w, xz or 0.0, wz or 0.0, [1, 1], g, [p, p, p, p], [st, st])
            assert got.shape == want.shape
            if _exact(x, xz or 0.0, w, wz or 0.0):
                assert np.array_equal(got, want), (n, c, h, oc, k, st, g, xz, wz)
                exact += 1
            else:
                _close(got, want)
    assert exact >= 12
    # asymmetric pads, a dilation, a fractional zero point (the f32 route) and an i8-coded weight tensor (the f32 route as well)
    x, w = _u8(rng, (2, 16, 9, 11)), _u8(rng, (8, 16, 3, 3))
    for pads, dil, xz in (([2, 0, 1, 3], [1, 1], 9.0), ([1, 1, 1, 1], [2, 1], 200.0), ([1, 2, 0, 1], [1, 1], 3.5)):
        got = K.conv_integer(x, Weight(w), np.array([xz], np.float32), np.array([128.0], np.float32), dil, 1, pads, [1, 1], ctx=ctx).numpy()
        want = O.conv_integer(x, w, xz, 128.0, dil, 1, pads, [1, 1])
        assert np.array_equal(got, want) if xz == int(xz) else _close(got, want) is None, (pads, dil, xz)
    wneg = w - np.float32(128)
    got = K.conv_integer(x, Weight(wneg), np.array([5.0], np.float32), None, [1, 1], 1, [1, 1, 1, 1], [1, 1], ctx=ctx).numpy()
    assert np.array_equal(got, O.conv_integer(x, wneg, 5.0, 0.0, [1, 1], 1, [1, 1, 1, 1], [1, 1]))
    # from_f32: dynamic quantisation of the activations inside the op (the zero point stays on the device)
    xf = (rng.standard_normal((2, 8, 10, 10)) * 3).astype(np.float32)
    w = _u8(rng, (12, 8, 3, 3))
    zw = np.array([128.0], np.float32)
    out, sc = K.conv_integer_from_f32(xf, Weight(w), zw, [1, 1], 1, [1, 1, 1, 1], [1, 1], ctx=ctx)
    s, z = npref.dql_params([xf])
    assert sc.numpy()[0] == s and z != 0
    want = O.conv_integer(npref.dql_quantize(xf, s, z), w, z, 128.0, [1, 1], 1, [1, 1, 1, 1], [1, 1])
    assert np.array_equal(out.numpy(), want)
    # multi: joint range over the sources, channel concatenation (5 + 3 channels: the second source starts off a 4-byte boundary), 1x1
    a = (rng.standard_normal((1, 5, 7, 7)) * 2).astype(np.float32)
    b = (rng.standard_normal((1, 3, 7, 7)) * 6 + 1).astype(np.float32)
    w1 = _u8(rng, (4, 8, 1, 1))
    out, sc = K.conv_integer_from_f32_multi([a, b], Weight(w1), zw, ctx=ctx)
    s, z = npref.dql_params([a, b])
    assert sc.numpy()[0] == s
    q = np.concatenate([npref.dql_quantize(a, s, z), npref.dql_quantize(b, s, z)], 1)
    want = O.conv_integer(q, w1, z, 128.0, [1, 1], 1, [0, 0, 0, 0], [1, 1])
    assert np.array_equal(out.numpy(), want)
    # fused_scale_bias(_silu): host float scale, and the device scale of conv_integer_from_f32 times a weight scale
    bias = rng.standard_normal(4).astype(np.float32)
    got = K.fused_scale_bias(out, sc, bias, scale_mul=0.02, ctx=ctx).numpy()
    eff = np.float32(s * np.float32(0.02))
    assert np.array_equal(got, npref.fused_scale_bias(want * 0 + out.numpy(), eff, bias))
    got = K.fused_scale_bias_silu(out, float(eff), bias, ctx=ctx).numpy()
    assert np.abs(got - npref.fused_scale_bias(out.numpy(), eff, bias, True)).max() <= 1e-5 * max(1.0, np.abs(got).max())
