"""Conv1d / Conv2d / ConvTranspose / LSTM / GRU.

CPU: the oracle is pinned on the reference's own test cases -- the conv1d unit tests (src/kernels/conv1d.rs:1621-1674),
the ref_conv2d / ref_gru_step cases of tests/regression_kernels.rs:75-252, 602-737 (same generated inputs, same
tolerances) -- and cross-checked against torch's CPU operators.  GPU: the HIP operators against the oracle through the
C ABI at the 1e-4 relative bar BASELINE.json states for f32 kernels.
"""
import numpy as np
import pytest

from oracle import pyoracle as O

RTOL = 1e-4


def _close(got, want, tol=RTOL, what=""):
    """north_star's bar, element by element: |got - want| <= tol * |want| + tol * rms(want).  The second term is the floor of a sum of
    K products in f32 -- an output that cancels to ~0 carries the round-off of its terms, which scale with the tensor's typical
    magnitude, not with its own -- and nothing else: no max(1, .) slack, no scaling by the tensor's LARGEST element (what this
    helper allowed until round 4).  Same form as tests/test_attention.py::close and tests/test_fullsize_graph.py."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not want.size:
        return
    floor = tol * float(np.sqrt(np.mean(np.square(want)))) + 1e-7
    bad = np.abs(got - want) > tol * np.abs(want) + floor
    assert not bad.any(), "%s: %d of %d elements outside %g (max abs diff %.3e, rms %.3e)" % (
        what, int(bad.sum()), want.size, tol, float(np.abs(got - want).max()), float(np.sqrt(np.mean(np.square(want)))))


def _seq(n, mul, add, mod=None):
    i = np.arange(n, dtype=np.int64)
    if mod:
        i = i % mod
    return (i.astype(np.float32) * np.float32(mul) + np.float32(add)).astype(np.float32)


def _ref_conv2d(x, w, b, stride, pad, group, relu):
    """plain-loop restatement of the in-test oracle ref_conv2d (tests/regression_kernels.rs:23-69), float64"""
    n, c, ih, iw = x.shape
    oc, icg, kh, kw = w.shape
    oh = (ih + 2 * pad - kh) // stride + 1
    ow = (iw + 2 * pad - kw) // stride + 1
    xp = np.zeros((n, c, ih + 2 * pad, iw + 2 * pad), np.float64)
    xp[:, :, pad:pad + ih, pad:pad + iw] = x
    out = np.zeros((n, oc, oh, ow), np.float64)
    ocg = oc // group
    for o in range(oc):
        g = o // ocg
        for y in range(oh):
            for xx in range(ow):
                win = xp[:, g * icg:(g + 1) * icg, y * stride:y * stride + kh, xx * stride:xx * stride + kw]
                out[:, o, y, xx] = (win * w[o].astype(np.float64)).sum(axis=(1, 2, 3))
        if b is not None:
            out[:, o] += b[o]
    return np.maximum(out, 0.0) if relu else out


CONV_KATS = [
    # (name, n, ic, oc, ih, iw, k, stride, pad, group, x(mul,add,mod), w(mul,add,mod), bias, relu, tol)
    ("3x3_s1_p1", 1, 2, 3, 8, 8, 3, 1, 1, 1, (0.1, -2.0, None), (0.05, -1.0, None), lambda oc: _seq(oc, 0.01, 0), True, 1e-3),
    ("3x3_s1_no_bias", 1, 1, 2, 6, 6, 3, 1, 1, 1, (0.3, -1.0, None), (0.2, -0.5, None), None, False, 1e-3),
    ("3x3_s2", 1, 3, 4, 16, 16, 3, 2, 1, 1, (0.01, -1.0, None), (0.03, 0.0, None),
     lambda oc: np.full(oc, 0.1, np.float32), True, 1e-3),
    ("1x1", 1, 16, 8, 4, 4, 1, 1, 0, 1, (0.1, 0.0, None), (0.01, 0.0, None), lambda oc: _seq(oc, 0.001, 0), True, 1e-3),
    ("pw_after_dw", 1, 32, 64, 4, 4, 1, 1, 0, 1, (0.05, 0.0, None), (0.02, -0.5, None), lambda oc: _seq(oc, 0.001, 0),
     True, 1e-2),
    # depthwise cases (regression_kernels.rs:133-210); upstream only asserts shape/finiteness there (its 4-channel
    # depthwise path is known-divergent, see the TODO at :165) -- the ONNX definition is the oracle here
    ("dw_4ch", 1, 4, 4, 10, 10, 3, 1, 1, 4, (0.2, -3.0, None), (0.1, -0.5, None), None, False, 1e-3),
    ("dw_64ch", 1, 64, 64, 8, 8, 3, 1, 1, 64, (0.07, -1.0, 97), (0.05, -0.3, 31), None, False, 1e-3),
    ("dw_128ch", 1, 128, 128, 6, 6, 3, 1, 1, 128, (0.03, 0.0, 73), (0.1, -0.5, 19), None, False, 1e-3),
]


def _kat_tensors(kat):
    name, n, ic, oc, ih, iw, k, stride, pad, group, xs, ws, bias, relu, tol = kat
    x = _seq(n * ic * ih * iw, *xs).reshape(n, ic, ih, iw)
    w = _seq(oc * (ic // group) * k * k, *ws).reshape(oc, ic // group, k, k)
    b = bias(oc) if bias else None
    return x, w, b


@pytest.mark.parametrize("kat", CONV_KATS, ids=[k[0] for k in CONV_KATS])
def test_oracle_conv2d_reference_cases(kat):
    name, n, ic, oc, ih, iw, k, stride, pad, group, xs, ws, bias, relu, tol = kat
    x, w, b = _kat_tensors(kat)
    want = _ref_conv2d(x, w, b, stride, pad, group, relu)
    got = O.conv2d(x, w, b, [1, 1], group, [pad] * 4, [stride, stride], "relu" if relu else None)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= tol  # assert_close of the reference is absolute, regression_kernels.rs:5-21


def test_oracle_conv1d_reference_unit_tests():
    # conv1d.rs:1621-1631 grouped, 1632-1643 simple, 1645-1674 k3 with padding
    y = O.conv1d(np.ones((1, 2, 3), np.float32), np.ones((2, 1, 1), np.float32), None, [1], 2, [0, 0], [1])
    assert y.shape == (1, 2, 3) and np.array_equal(y.ravel(), np.ones(6))
    y = O.conv1d(np.array([1, 2, 3], np.float32).reshape(1, 1, 3), np.ones((1, 1, 2), np.float32), None, [1], 1, [0, 0], [1])
    assert y.shape == (1, 1, 2) and np.array_equal(y.ravel(), [3, 5])
    y = O.conv1d(np.arange(10, dtype=np.float32).reshape(1, 1, 10), np.ones((1, 1, 3), np.float32), None, [1], 1, [1, 1], [1])
    assert y.shape == (1, 1, 10) and (y[0, 0, 0], y[0, 0, 1], y[0, 0, 5], y[0, 0, 9]) == (1.0, 3.0, 15.0, 17.0)


def test_oracle_conv_vs_torch():
    import torch
    import torch.nn.functional as F
    rng = np.random.default_rng(5)
    x = rng.standard_normal((2, 6, 11, 9)).astype(np.float32)
    w = rng.standard_normal((8, 3, 3, 2)).astype(np.float32)
    b = rng.standard_normal(8).astype(np.float32)
    want = F.conv2d(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), stride=(2, 1), padding=(1, 2),
                    dilation=(1, 2), groups=2).numpy()
    _close(O.conv2d(x, w, b, [1, 2], 2, [1, 2, 1, 2], [2, 1]), want, 1e-5, "conv2d")
    _close(O.conv2d(x, w, b, [1, 2], 2, [1, 2, 1, 2], [2, 1], "silu"), F.silu(torch.from_numpy(want)).numpy(), 1e-5, "silu")
    wt = rng.standard_normal((6, 4, 3, 3)).astype(np.float32)
    bt = rng.standard_normal(4).astype(np.float32)
    want = F.conv_transpose2d(torch.from_numpy(x), torch.from_numpy(wt), torch.from_numpy(bt), stride=2, padding=1).numpy()
    _close(O.conv_transpose(x, wt, bt, [1, 1], 1, [1, 1, 1, 1], [2, 2]), want, 1e-5, "conv_transpose")
    x1 = rng.standard_normal((2, 4, 50)).astype(np.float32)
    w1 = rng.standard_normal((6, 2, 5)).astype(np.float32)
    want = F.conv1d(torch.from_numpy(x1), torch.from_numpy(w1), None, stride=2, padding=2, dilation=2, groups=2).numpy()
    _close(O.conv1d(x1, w1, None, [2], 2, [2, 2], [2]), want, 1e-5, "conv1d")


def _ref_gru(x, w, r, bw, br, hs):
    """plain restatement of ref_gru_step (tests/regression_kernels.rs:602-633), float64 sigmoid/tanh"""
    h = np.zeros(hs)
    ys = []
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    for xt in x:
        wc, rc = w.astype(np.float64) @ xt, r.astype(np.float64) @ h
        z = sig(wc[:hs] + rc[:hs] + bw[:hs] + br[:hs])
        rg = sig(wc[hs:2 * hs] + rc[hs:2 * hs] + bw[hs:2 * hs] + br[hs:2 * hs])
        hg = np.tanh(wc[2 * hs:] + bw[2 * hs:] + rg * (rc[2 * hs:] + br[2 * hs:]))
        h = (1.0 - z) * hg + z * h
        ys.append(h.copy())
    return np.array(ys), h


GRU_KATS = [
    # (name, T, I, H, x, w(mul,add), r(mul,add), bias, lbr, tol)   regression_kernels.rs:635-737
    ("single_step", 1, 4, 8, np.array([0.1, 0.2, -0.1, 0.3], np.float32), (0.01, -0.1), (0.02, -0.2),
     lambda n: _seq(n, 0.005, -0.05), False, 1e-4),
    ("multi_step", 5, 3, 6, ((np.arange(15) * 7 + 3) % 20).astype(np.float32) * np.float32(0.1) - np.float32(0.5),
     (0.03, -0.2), (0.01, -0.1), lambda n: np.full(n, 0.1, np.float32), False, 1e-3),
    ("linear_before_reset", 3, 4, 8, _seq(12, 0.15, -0.3), (0.01, 0.0), (0.02, -0.1),
     lambda n: np.full(n, 0.05, np.float32), True, 1e-3),
    ("no_bias", 2, 3, 4, np.array([0.5, -0.3, 0.1, -0.2, 0.4, 0.6], np.float32), (0.05, 0.0), (0.03, 0.0), None, False, 1e-4),
]


def _gru_tensors(kat):
    name, T, I, H, x, ws, rs, bias, lbr, tol = kat
    x = np.asarray(x, np.float32).reshape(T, 1, I)
    w = _seq(3 * H * I, *ws).reshape(1, 3 * H, I)
    r = _seq(3 * H * H, *rs).reshape(1, 3 * H, H)
    b = bias(6 * H).reshape(1, 6 * H) if bias else None
    return x, w, r, b


@pytest.mark.parametrize("kat", GRU_KATS, ids=[k[0] for k in GRU_KATS])
def test_oracle_gru_reference_cases(kat):
    name, T, I, H, _, _, _, _, lbr, tol = kat
    x, w, r, b = _gru_tensors(kat)
    bw = b[0, :3 * H] if b is not None else np.zeros(3 * H)
    br = b[0, 3 * H:] if b is not None else np.zeros(3 * H)
    yref, href = _ref_gru(x[:, 0, :], w[0], r[0], bw, br, H)
    y, h = O.gru(x, w, r, b)
    assert y.shape == (T, 1, 1, H) and h.shape == (1, 1, H)
    assert np.abs(y.reshape(T, H) - yref).max() <= tol and np.abs(h.ravel() - href).max() <= tol


def test_oracle_rnn_vs_torch():
    import torch
    rng = np.random.default_rng(3)
    T, I, H = 7, 12, 20  # H = 20: two 8-wide polynomial blocks + a 4-element libm tail
    x = rng.standard_normal((T, 1, I)).astype(np.float32)
    h0 = rng.standard_normal((1, 1, H)).astype(np.float32) * 0.3
    c0 = rng.standard_normal((1, 1, H)).astype(np.float32) * 0.3
    # LSTM: lele's gate order is i, o, f, c (ONNX); torch's is i, f, g(c), o
    w = rng.standard_normal((1, 4 * H, I)).astype(np.float32) * 0.3
    r = rng.standard_normal((1, 4 * H, H)).astype(np.float32) * 0.3
    b = rng.standard_normal((1, 8 * H)).astype(np.float32) * 0.3
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(3 * H, 4 * H), np.arange(H, 2 * H)])
    m = torch.nn.LSTM(I, H)
    with torch.no_grad():
        m.weight_ih_l0.copy_(torch.from_numpy(w[0][perm]))
        m.weight_hh_l0.copy_(torch.from_numpy(r[0][perm]))
        m.bias_ih_l0.copy_(torch.from_numpy(b[0, :4 * H][perm]))
        m.bias_hh_l0.copy_(torch.from_numpy(b[0, 4 * H:][perm]))
        ty, (th, tc) = m(torch.from_numpy(x), (torch.from_numpy(h0), torch.from_numpy(c0)))
    y, h, c = O.lstm(x, w, r, b, h0, c0)
    _close(y.reshape(T, H), ty.numpy().reshape(T, H), 1e-5, "lstm y")
    _close(h.ravel(), th.numpy().ravel(), 1e-5, "lstm h")
    _close(c.ravel(), tc.numpy().ravel(), 1e-5, "lstm c")
    # GRU: lele z, r, h; torch r, z, n (and torch is the linear_before_reset = 1 form)
    w = rng.standard_normal((1, 3 * H, I)).astype(np.float32) * 0.3
    r = rng.standard_normal((1, 3 * H, H)).astype(np.float32) * 0.3
    b = rng.standard_normal((1, 6 * H)).astype(np.float32) * 0.3
    perm = np.concatenate([np.arange(H, 2 * H), np.arange(0, H), np.arange(2 * H, 3 * H)])
    m = torch.nn.GRU(I, H)
    with torch.no_grad():
        m.weight_ih_l0.copy_(torch.from_numpy(w[0][perm]))
        m.weight_hh_l0.copy_(torch.from_numpy(r[0][perm]))
        m.bias_ih_l0.copy_(torch.from_numpy(b[0, :3 * H][perm]))
        m.bias_hh_l0.copy_(torch.from_numpy(b[0, 3 * H:][perm]))
        ty, th = m(torch.from_numpy(x), torch.from_numpy(h0))
    y, h = O.gru(x, w, r, b, h0)
    _close(y.reshape(T, H), ty.numpy().reshape(T, H), 1e-5, "gru y")
    _close(h.ravel(), th.numpy().ravel(), 1e-5, "gru h")


# --------------------------------------------------------------------------------------------------- device
@pytest.mark.gpu
def test_conv_stats_count_calls_and_macs(ctx):
    """reset_conv_stats / print_conv_stats (conv2d.rs:75,101; the YOLO example calls them around its timed loop)"""
    from lele_amd import kernels as K
    rng = np.random.default_rng(0)
    x, w = rng.standard_normal((2, 3, 10, 12)).astype(np.float32), rng.standard_normal((5, 3, 3, 3)).astype(np.float32)
    K.reset_conv_stats(ctx=ctx)
    assert K.conv_stats(ctx=ctx) == (0, 0)
    y = K.conv2d(x, w, None, (), 1, (1, 1, 1, 1), (1, 1), ctx=ctx)
    K.conv2d_silu(x, w, None, (), 1, (0, 0, 0, 0), (2, 2), ctx=ctx)
    assert y.shape == (2, 5, 10, 12)
    assert K.conv_stats(ctx=ctx) == (2, 2 * 5 * 10 * 12 * 27 + 2 * 5 * 4 * 5 * 27)
    K.print_conv_stats(ctx=ctx)
    K.reset_conv_stats(ctx=ctx)
    assert K.conv_stats(ctx=ctx) == (0, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("kat", CONV_KATS, ids=[k[0] for k in CONV_KATS])
def test_device_conv2d_reference_cases(ctx, kat):
    from lele_amd import kernels as K
    name, n, ic, oc, ih, iw, k, stride, pad, group, xs, ws, bias, relu, tol = kat
    x, w, b = _kat_tensors(kat)
    got = K.conv2d_fused(x, w, b, [1, 1], group, [pad] * 4, [stride, stride], relu, ctx=ctx).numpy()
    want = O.conv2d(x, w, b, [1, 1], group, [pad] * 4, [stride, stride], "relu" if relu else None)
    _close(got, want, RTOL, name)


CONV_SHAPES = [
    # n, c, h, w, oc, kh, kw, group, pads, strides, dilations, act
    (1, 3, 64, 64, 16, 3, 3, 1, [1, 1, 1, 1], [2, 2], [1, 1], "silu"),       # yolo stem shape class
    (2, 32, 40, 40, 64, 3, 3, 1, [1, 1, 1, 1], [1, 1], [1, 1], "silu"),
    (1, 64, 20, 20, 64, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "relu"),
    (1, 16, 33, 29, 24, 5, 3, 2, [2, 1, 0, 3], [2, 1], [1, 2], None),         # ragged everything
    (1, 128, 17, 17, 128, 3, 3, 128, [1, 1, 1, 1], [1, 1], [1, 1], "silu"),   # depthwise + activation
    (3, 8, 9, 9, 16, 3, 3, 8, [1, 1], [1], [1], "relu"),                       # channel multiplier 2, short attr forms
    (1, 4, 7, 5, 6, 7, 5, 1, [], [], [], None),                                # kernel == input -> 1x1 output
    (1, 96, 12, 12, 200, 3, 3, 1, [1, 1, 1, 1], [1, 1], [1, 1], None),         # OC > 128 tile, K = 864
    # few output channels over a batch (>= 512 tiles of 8 x 32 pixels): the direct 3 x 3 kernel -- stride 1 / 2, ragged maps,
    # asymmetric pads, OC not a multiple of 8, more input channels than one LDS pass holds
    (48, 16, 48, 48, 8, 3, 3, 1, [1, 1, 1, 1], [1, 1], [1, 1], "silu"),
    (64, 3, 70, 66, 16, 3, 3, 1, [1, 1, 1, 1], [2, 2], [1, 1], "silu"),
    (90, 8, 37, 45, 13, 3, 3, 1, [0, 2, 1, 0], [1, 1], [1, 1], "relu"),
    (176, 64, 40, 36, 5, 3, 3, 1, [1, 0, 1, 2], [2, 2], [1, 1], None),
    # IC % 16 == 0, OC % 64 == 0, stride 1, >= one tile per CU: the window-once split-bf16 MFMA kernel -- 16-byte stores (ow % 4 == 0,
    # partial tiles in x and y), the scalar store path (ow % 4 != 0), asymmetric pads, one and several channel chunks, odd chunk count
    (32, 32, 40, 44, 128, 3, 3, 1, [1, 1, 1, 1], [1, 1], [1, 1], "silu"),
    (40, 16, 30, 37, 64, 3, 3, 1, [1, 0, 1, 2], [1, 1], [1, 1], "relu"),
    (32, 48, 33, 64, 64, 3, 3, 1, [0, 1, 2, 1], [1, 1], [1, 1], None),
    # stride 2 over a batch (the de-interleaved window kernel), even and odd input sizes
    (32, 32, 80, 80, 64, 3, 3, 1, [1, 1, 1, 1], [2, 2], [1, 1], "silu"),
    (40, 16, 41, 77, 96, 3, 3, 1, [0, 1, 1, 0], [2, 2], [1, 1], "relu"),
    # the same kernel in its other forms: blocks of 32 output channels with a ragged last block (OC = 80), and 1 x 1 on a large plane
    (40, 32, 40, 40, 80, 3, 3, 1, [1, 1, 1, 1], [1, 1], [1, 1], "silu"),
    (32, 48, 80, 80, 64, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "silu"),
    (32, 64, 80, 84, 24, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], None),
    (48, 96, 40, 44, 100, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "silu"),   # 64 < OC <= 128 (the tiled GEMM: an eight-consumer-wave form of the window kernel measured no faster)
    # 1 x 1 over a batch: several K steps, odd IC, OC off the 32-channel tile, planes that are / are not a multiple of 4 (16-byte /
    # scalar stores of the tiled kernel's epilogue) and of 32
    (96, 48, 40, 40, 64, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "silu"),
    (128, 33, 36, 30, 40, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "relu"),
    (168, 16, 27, 29, 96, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], None),
    (64, 130, 48, 40, 200, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "silu"),
    # the tile shapes the host picks per map (pick_win_tile: row-major strips of 32 positions, not rows of 32): 20 x 12 on a 20-wide
    # map, 16 x 16 on an 80-wide one, 20 x 6 / 40 x 3 at stride 2, a 1 x 1 plane as one row of 256-position tiles with a ragged last
    # tile (16-byte stores) and with an odd plane (scalar stores)
    (64, 32, 20, 20, 64, 3, 3, 1, [1, 1, 1, 1], [1, 1], [1, 1], "silu"),
    (16, 32, 80, 80, 64, 3, 3, 1, [1, 1, 1, 1], [1, 1], [1, 1], "relu"),
    (128, 32, 40, 40, 64, 3, 3, 1, [1, 1, 1, 1], [2, 2], [1, 1], "silu"),
    (48, 16, 80, 80, 32, 3, 3, 1, [1, 1, 1, 1], [2, 2], [1, 1], None),
    (32, 48, 50, 36, 64, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "silu"),
    (32, 48, 45, 41, 64, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "relu"),
    # 1 x 1 with two or more blocks of output channels: 128-channel blocks over tiles of 128 positions (four waves x 32 channels):
    # whole blocks, a padded last block (250 of 256), a ragged last tile, an odd plane (scalar stores)
    (64, 32, 40, 40, 128, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "silu"),
    (64, 48, 36, 30, 250, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], "relu"),
    (72, 16, 27, 29, 128, 1, 1, 1, [0, 0, 0, 0], [1, 1], [1, 1], None),
    # stride 2 (3 x 3) with two and four blocks of output channels, 250 of 256 on a ragged map
    (128, 32, 40, 40, 128, 3, 3, 1, [1, 1, 1, 1], [2, 2], [1, 1], "silu"),
    (96, 16, 50, 46, 250, 3, 3, 1, [1, 1, 1, 1], [2, 2], [1, 1], "relu"),
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", CONV_SHAPES, ids=[str(i) for i in range(len(CONV_SHAPES))])
def test_device_conv2d_vs_oracle(ctx, shape):
    from lele_amd import kernels as K
    n, c, h, w_, oc, kh, kw, g, pads, strides, dil, act = shape
    rng = np.random.default_rng(hash(shape[:8]) & 0xffff)
    x = rng.standard_normal((n, c, h, w_)).astype(np.float32)
    w = (rng.standard_normal((oc, c // g, kh, kw)) * 0.2).astype(np.float32)
    b = rng.standard_normal(oc).astype(np.float32)
    fn = {"silu": K.conv2d_silu, "relu": lambda *a, **k: K.conv2d_fused(*a, relu=True, **k), None: K.conv2d}[act]
    got = fn(x, w, b, dil, g, pads, strides, ctx=ctx).numpy()
    _close(got, O.conv2d(x, w, b, dil, g, pads, strides, act), RTOL, str(shape))
    got = fn(x, w, None, dil, g, pads, strides, ctx=ctx).numpy()
    _close(got, O.conv2d(x, w, None, dil, g, pads, strides, act), RTOL, str(shape) + " no bias")


@pytest.mark.gpu
@pytest.mark.parametrize("n,c,hw", [(2, 32, 40), (64, 32, 20), (64, 8, 24), (1, 16, 17)], ids=["tiled", "window", "narrow", "ragged"])
def test_conv_epilogue_silu_default_and_exact_forms(ctx, monkeypatch, n, c, hw):
    """The convolution epilogue's SiLU alone: a 1 x 1 convolution with identity weights hands every input value to the activation
    unchanged (x * 1 + zeros is exact on every route, split-bf16 included).  Default form (v_exp_f32 / v_rcp_f32): within 1e-5
    relative + 1e-7 of the reference's (polynomial body, libm tail) over the whole range, the tails of the range included.
    LELE_HIP_CONV_SILU_EXACT=1: the replica, bit for bit where the reference's vector form applies."""
    from lele_amd import kernels as K
    rng = np.random.default_rng(n * 1000 + c)
    count = n * c * hw * hw
    vals = np.concatenate([np.linspace(-110.0, 110.0, count // 4), rng.standard_normal(count // 4) * 3.0,
                           rng.standard_normal(count // 4) * 30.0,
                           np.sign(rng.standard_normal(count - 3 * (count // 4))) * np.exp(rng.uniform(-60.0, 4.0, count - 3 * (count // 4)))])   # (below 2^-110 a value's third bf16 piece is denormal)
    x = rng.permutation(vals).astype(np.float32).reshape(n, c, hw, hw)
    w = np.eye(c, dtype=np.float32).reshape(c, c, 1, 1)
    want = O.conv2d(x, w, None, (), 1, (0, 0, 0, 0), (1, 1), "silu")
    monkeypatch.delenv("LELE_HIP_CONV_SILU_EXACT", raising=False)
    got = K.conv2d_silu(x, w, None, (), 1, (0, 0, 0, 0), (1, 1), ctx=ctx).numpy()
    err = np.abs(got.astype(np.float64) - want)
    assert np.all(err <= 1e-5 * np.abs(want) + 1e-7), (float(err.max()), float((err / (np.abs(want) + 1e-30)).max()))
    monkeypatch.setenv("LELE_HIP_CONV_SILU_EXACT", "1")
    exact = K.conv2d_silu(x, w, None, (), 1, (0, 0, 0, 0), (1, 1), ctx=ctx).numpy()
    body = (hw * hw) & ~7   # the last 0-7 positions of a plane take the reference's scalar form: libm's expf, the device's own there
    ef, wf = exact.reshape(n, c, -1), want.reshape(n, c, -1)
    assert np.array_equal(ef[..., :body], wf[..., :body])
    assert np.all(np.abs(ef[..., body:].astype(np.float64) - wf[..., body:]) <= 4e-7 * np.abs(wf[..., body:]) + 1e-30)
    bias = rng.standard_normal(c).astype(np.float32)   # and with a bias / through the 3 x 3 routes the replica stays inside the usual bar
    w3 = (rng.standard_normal((c, c, 3, 3)) * 0.2).astype(np.float32)
    _close(K.conv2d_silu(x, w3, bias, (), 1, (1, 1, 1, 1), (1, 1), ctx=ctx).numpy(), O.conv2d(x, w3, bias, (), 1, (1, 1, 1, 1), (1, 1), "silu"), RTOL, "exact 3x3")


@pytest.mark.gpu
def test_batched_convolution_kernels_agree_with_single_image_calls(ctx):
    """Over a batch the 3 x 3 / 1 x 1 convolutions take their own kernels (window-once split-bf16 MFMA, direct small-channel);
    one image at a time takes the implicit GEMM.  Random geometries -- channels, ragged maps, asymmetric pads, stride 1 / 2, every
    activation -- must agree image by image within the 1e-4 bar (the two are different summation orders of the same exact products)."""
    from lele_amd import kernels as K
    rng = np.random.default_rng(2024)
    acts = {"silu": K.conv2d_silu, "relu": lambda *a, **k: K.conv2d_fused(*a, relu=True, **k), None: K.conv2d}
    for trial in range(14):
        k = int(rng.choice([1, 3, 3]))
        c = int(rng.choice([3, 8, 16, 32, 48, 64, 96])) if k == 3 else int(rng.choice([48, 64, 96, 128]))
        oc = int(rng.choice([5, 8, 13, 16, 24, 32, 40, 64, 80, 128])) if k == 3 else int(rng.choice([24, 32, 64]))
        stride = int(rng.choice([1, 1, 2])) if k == 3 else 1
        h, w_ = (int(rng.integers(40, 72)), int(rng.integers(40, 100))) if k == 3 else (int(rng.integers(80, 90)), int(rng.integers(80, 96)))
        pads = [int(v) for v in rng.integers(0, 3, 4)] if k == 3 else [0, 0, 0, 0]
        oh = (h + pads[0] + pads[2] - k) // stride + 1
        ow = (w_ + pads[1] + pads[3] - k) // stride + 1
        tiles = ((ow + 31) // 32) * ((oh + 7) // 8)
        n = max(2, -(-600 // tiles))  # enough 8 x 32 tiles for the batched kernels' thresholds
        act = [None, "relu", "silu"][trial % 3]
        x = rng.standard_normal((n, c, h, w_)).astype(np.float32)
        w = (rng.standard_normal((oc, c, k, k)) * (2.0 / (c * k * k)) ** 0.5).astype(np.float32)
        b = rng.standard_normal(oc).astype(np.float32) if trial % 4 else None
        got = acts[act](x, w, b, [1, 1], 1, pads, [stride, stride], ctx=ctx).numpy()
        assert got.shape == (n, oc, oh, ow) and np.isfinite(got).all()
        for i in sorted(set(int(v) for v in rng.integers(0, n, 3))):
            one = acts[act](x[i:i + 1], w, b, [1, 1], 1, pads, [stride, stride], ctx=ctx).numpy()
            _close(got[i:i + 1], one, RTOL, "trial %d: [%d,%d,%d,%d] -> %d, k%d s%d pads %s %s, image %d" % (trial, n, c, h, w_, oc, k, stride, pads, act, i))


@pytest.mark.gpu
def test_window_kernels_are_deterministic_and_the_residual_is_the_conv_plus_add(ctx):
    """The window kernels hand LDS stages between loader and multiplying waves at raw s_barriers.  Until round 6 nothing waited for a
    wave's own ds_writes in front of such a barrier: a parked chunk could still be on its way into LDS when the stage changed hands --
    it showed as one tile 3e-6 off in one run of two (256 -> 64 channels 1 x 1 at 80 x 80 x 64) once the epilogue's timing moved.  Deep
    1 x 1 layers on large planes, several runs each: conv2d_res must be conv2d_silu followed by add BIT FOR BIT, and a second run of the
    same call must reproduce the first (both are exact statements about one summation order)."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(6)
    for n, c, oc, k, hw in ((64, 256, 64, 1, 80), (64, 96, 64, 1, 80), (64, 64, 64, 3, 40), (32, 128, 128, 1, 40)):
        x = ctx.buf().upload(rng.standard_normal((n, c, hw, hw)).astype(np.float32))
        r = ctx.buf().upload(rng.standard_normal((n, oc, hw, hw)).astype(np.float32))
        w = Weight((rng.standard_normal((oc, c, k, k)) * 0.1).astype(np.float32))
        b = Weight(rng.standard_normal(oc).astype(np.float32))
        args = ([1, 1], 1, [k // 2] * 4, [1, 1])
        first = None
        for rep in range(3):
            y = K.conv2d_silu(x, w, b, *args, out=ctx.buf(), ctx=ctx)
            want = K.add(y, r, out=ctx.buf(), ctx=ctx).numpy()
            got = K.conv2d_res(x, w, b, r, *args, act=2, out=ctx.buf(), ctx=ctx).numpy()
            assert np.array_equal(want, got), "conv2d_res != conv2d_silu + add: %d -> %d k%d at %d x %d x %d, run %d: %d elements, max |d| %.3g" % (
                c, oc, k, hw, hw, n, rep, int((want != got).sum()), float(np.abs(want - got).max()))
            if first is None:
                first = got
            assert np.array_equal(first, got), "run %d differs from run 0: %d -> %d k%d" % (rep, c, oc, k)


@pytest.mark.gpu
def test_device_conv1d(ctx):
    from lele_amd import kernels as K
    y = K.conv1d(np.ones((1, 2, 3), np.float32), np.ones((2, 1, 1), np.float32), None, [1], 2, [0, 0], [1], ctx=ctx)
    assert y.shape == (1, 2, 3) and np.array_equal(y.numpy().ravel(), np.ones(6))
    y = K.conv1d(np.arange(10, dtype=np.float32).reshape(1, 1, 10), np.ones((1, 1, 3), np.float32), None, [1], 1, [1, 1],
                 [1], ctx=ctx).numpy()
    assert y.shape == (1, 1, 10) and (y[0, 0, 0], y[0, 0, 1], y[0, 0, 5], y[0, 0, 9]) == (1.0, 3.0, 15.0, 17.0)
    rng = np.random.default_rng(8)
    cases = [
        ((1, 1, 4000), (258, 1, 256), 1, [], [128], [], False),      # STFT-as-conv of the VAD model (conv1d.rs:899)
        ((2, 64, 301), (128, 64, 3), 1, [1, 1], [2], [1], True),
        ((1, 80, 500), (80, 1, 5), 80, [2, 2], [1], [1], False),     # depthwise 1-D
        ((1, 16, 77), (32, 4, 7), 4, [9, 3], [3], [2], True),
        ((3, 200), (4, 1, 9), 1, [4], [1], [1], False),              # rank-2 input, single pad value (right pad 0)
    ]
    for xs, ws, g, pads, strides, dil, relu in cases:
        x = rng.standard_normal(xs).astype(np.float32)
        w = (rng.standard_normal(ws) * 0.2).astype(np.float32)
        b = rng.standard_normal(ws[0]).astype(np.float32)
        got = K.conv1d_fused(x, w, b, dil, g, pads, strides, relu, ctx=ctx).numpy()
        _close(got, O.conv1d(x, w, b, dil, g, pads, strides, relu), RTOL, str((xs, ws)))


@pytest.mark.gpu
def test_device_conv_transpose(ctx):
    import lele_amd
    from lele_amd import kernels as K
    rng = np.random.default_rng(9)
    for xs, ws, pads, strides, dil in [((1, 8, 10, 10), (8, 4, 2, 2), [], [2, 2], []),
                                       ((2, 6, 7, 9), (6, 5, 3, 3), [1, 1, 1, 1], [2, 2], [1, 1]),
                                       ((1, 3, 5, 6), (3, 7, 4, 3), [1, 0, 2, 1], [3, 2], [2, 1]),
                                       ((1, 16, 20, 20), (16, 16, 3, 3), [1, 1, 1, 1], [1, 1], [1, 1]),
                                       ((1, 4, 5, 5), (4, 3, 1, 1), [], [2, 2], []),          # stride > kernel: bias-only phases
                                       ((2, 64, 12, 12), (64, 40, 2, 2), [], [2, 2], []),     # the YOLO neck upsampling shape class
                                       ((1, 5, 6, 7), (5, 6, 3, 5), [0, 2, 1, 0], [2, 3], [3, 2]),
                                       # kernel = stride = 2 over a batch (one GEMM with the scattering quad-store epilogue): ragged tiles, OC not a
                                       # multiple of 16
                                       ((80, 16, 40, 40), (16, 12, 2, 2), [], [2, 2], []),
                                       ((40, 32, 36, 64), (32, 20, 2, 2), [], [2, 2], [])]:
        x = rng.standard_normal(xs).astype(np.float32)
        w = (rng.standard_normal(ws) * 0.2).astype(np.float32)
        b = rng.standard_normal(ws[1]).astype(np.float32)
        got = K.conv_transpose(x, w, b, dil, 1, pads, strides, ctx=ctx).numpy()
        _close(got, O.conv_transpose(x, w, b, dil, 1, pads, strides), RTOL, str((xs, ws)))
        if xs[0] >= 40:   # ... and without a bias
            _close(K.conv_transpose(x, w, None, dil, 1, pads, strides, ctx=ctx).numpy(), O.conv_transpose(x, w, None, dil, 1, pads, strides), RTOL, str((xs, ws)))
    with pytest.raises(lele_amd.LeleError, match="group > 1 not supported"):
        K.conv_transpose(np.zeros((1, 4, 3, 3), np.float32), np.zeros((4, 2, 2, 2), np.float32), None, [], 2, [], [], ctx=ctx)


@pytest.mark.gpu
@pytest.mark.parametrize("kat", GRU_KATS, ids=[k[0] for k in GRU_KATS])
def test_device_gru_reference_cases(ctx, kat):
    from lele_amd import kernels as K
    name, T, I, H, _, _, _, _, lbr, tol = kat
    x, w, r, b = _gru_tensors(kat)
    y, h = K.gru(x, w, r, b, None, lbr, ctx=ctx)
    yo, ho = O.gru(x, w, r, b)
    assert y.shape == (T, 1, 1, H) and h.shape == (1, 1, H)
    _close(y.numpy(), yo, RTOL, name + " y")
    _close(h.numpy(), ho, RTOL, name + " h")


@pytest.mark.gpu
@pytest.mark.parametrize("T,I,H", [(1, 128, 128), (50, 64, 128), (9, 13, 20), (4, 300, 517), (3, 7, 1)])
def test_device_lstm_gru_vs_oracle(ctx, T, I, H):
    import lele_amd
    from lele_amd import kernels as K
    rng = np.random.default_rng(T * 1000 + H)
    x = rng.standard_normal((T, 1, I)).astype(np.float32)
    sc = np.float32(1.0 / np.sqrt(max(I, H)))
    h0 = (rng.standard_normal((1, 1, H)) * 0.5).astype(np.float32)
    c0 = (rng.standard_normal((1, 1, H)) * 0.5).astype(np.float32)
    w = (rng.standard_normal((1, 4 * H, I)) * sc).astype(np.float32)
    r = (rng.standard_normal((1, 4 * H, H)) * sc).astype(np.float32)
    b = (rng.standard_normal((1, 8 * H)) * 0.2).astype(np.float32)
    for bias, hh, cc in ((b, h0, c0), (None, None, None)):
        y, h, c = K.lstm(x, w, r, bias, None, hh, cc, ctx=ctx)
        yo, ho, co = O.lstm(x, w, r, bias, hh, cc)
        assert y.shape == (T, 1, 1, H) and h.shape == (1, 1, H) and c.shape == (1, 1, H)
        _close(y.numpy(), yo, RTOL, "lstm y")
        _close(h.numpy(), ho, RTOL, "lstm h")
        _close(c.numpy(), co, RTOL, "lstm c")
    w, r, b = w[:, :3 * H], r[:, :3 * H], b[:, :6 * H]
    for bias, hh in ((b, h0), (None, None)):
        y, h = K.gru(x, w, r, bias, hh, False, ctx=ctx)
        yo, ho = O.gru(x, w, r, bias, hh)
        _close(y.numpy(), yo, RTOL, "gru y")
        _close(h.numpy(), ho, RTOL, "gru h")
    with pytest.raises(lele_amd.LeleError, match="Only batch_size=1 supported"):
        K.lstm(np.zeros((2, 2, I), np.float32), w_pad(w, 4 * H, I), r_pad(r, 4 * H, H), ctx=ctx)


def w_pad(w, g, i):
    return np.zeros((1, g, i), np.float32)


def r_pad(r, g, h):
    return np.zeros((1, g, h), np.float32)


@pytest.mark.gpu
def test_device_rnn_state_in_place(ctx):
    """the final state may be stored where the initial state was read from (out_h / out_c = the buffers of initial_h / initial_c):
    the kernel takes the initial state into LDS before its first step and stores the final state after the last one.  A
    streaming caller (Silero, chunk after chunk) keeps the state on the device this way, with no copy in between."""
    from lele_amd import kernels as K
    from lele_amd.tensor import TensorView
    rng = np.random.default_rng(77)
    T, I, H = 3, 24, 128
    sc = np.float32(1.0 / np.sqrt(H))
    w, r = ((rng.standard_normal((1, 4 * H, n)) * sc).astype(np.float32) for n in (I, H))
    b = (rng.standard_normal((1, 8 * H)) * 0.2).astype(np.float32)
    xs = [rng.standard_normal((T, 1, I)).astype(np.float32) for _ in range(4)]
    h = c = np.zeros((1, 1, H), np.float32)
    want = []
    for x in xs:                                             # separate state buffers, state round-tripped through the host
        y, hn, cn = K.lstm(x, w, r, b, None, h, c, ctx=ctx)
        h, c = hn.numpy().copy(), cn.numpy().copy()
        want.append((y.numpy().copy(), h, c))
    hb, cb, yb = ctx.buf(), ctx.buf(), ctx.buf()
    hv, cv = TensorView(hb.upload(np.zeros((1, 1, H), np.float32))), TensorView(cb.upload(np.zeros((1, 1, H), np.float32)))
    for x, (yw, hw, cw) in zip(xs, want):                    # in place: initial_h / initial_c and out_h / out_c share buffers
        y, hn, cn = K.lstm(x, w, r, b, None, hv, cv, outs=[yb, hb, cb], ctx=ctx)
        assert np.array_equal(y.raw().numpy(), yw) and np.array_equal(hn.raw().numpy(), hw) and np.array_equal(cn.raw().numpy(), cw)
    w3, r3, b3 = w[:, :3 * H], r[:, :3 * H], b[:, :6 * H]
    h = np.zeros((1, 1, H), np.float32)
    hv = TensorView(hb.upload(h))
    for x in xs:
        y, hn = K.gru(x, w3, r3, b3, h, False, ctx=ctx)
        h = hn.numpy().copy()
        y2, hn2 = K.gru(x, w3, r3, b3, hv, False, outs=[yb, hb], ctx=ctx)
        assert np.array_equal(y2.raw().numpy(), y.numpy()) and np.array_equal(hn2.raw().numpy(), h)


THIN_CONVS = [
    # Silero-shaped: c_in, c_out, k, stride, pad, length -> 1..4 output positions (the matrix-vector path of gemm_core.h)
    (1, 258, 256, 128, 0, 576), (129, 128, 3, 1, 1, 3), (128, 64, 3, 2, 1, 3), (64, 64, 3, 2, 1, 2), (64, 128, 3, 1, 1, 1),
    (128, 1, 1, 1, 0, 1), (7, 13, 5, 1, 2, 4), (24, 40, 3, 1, 0, 3), (16, 9, 4, 3, 0, 13),
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", THIN_CONVS, ids=[str(i) for i in range(len(THIN_CONVS))])
def test_device_thin_convolutions_vs_oracle(ctx, shape):
    from lele_amd import kernels as K
    ci, co, k, s, p, n = shape
    rng = np.random.default_rng(co * 100 + n)
    for batch, group in ((1, 1), (3, 1)) + (((2, 2),) if ci % 2 == 0 and co % 2 == 0 else ()):
        x = rng.standard_normal((batch, ci, n)).astype(np.float32)
        w = (rng.standard_normal((co, ci // group, k)) / np.sqrt(ci * k)).astype(np.float32)
        b = rng.standard_normal(co).astype(np.float32)
        for bias, relu in ((b, True), (None, False)):
            got = K.conv1d_fused(x, w, bias, [1], group, [p, p], [s], relu, ctx=ctx).numpy()
            want = O.conv1d(x, w, bias, [1], group, [p, p], [s], relu)
            assert got.shape[-1] <= 4, got.shape
            _close(got, want, RTOL, "thin conv1d %s batch %d group %d" % (shape, batch, group))
    # 2-D: a 2x2 (and 1x3) output plane
    x = rng.standard_normal((2, ci, 4, 5)).astype(np.float32)
    w = (rng.standard_normal((co, ci, 3, 3)) / np.sqrt(ci * 9)).astype(np.float32)
    for strides, pads in (([1, 2], [0, 0, 0, 0]), ([2, 2], [0, 1, 0, 0])):
        got = K.conv2d(x, w, None, [1, 1], 1, pads, strides, ctx=ctx).numpy()
        assert got.shape[2] * got.shape[3] <= 4, got.shape
        _close(got, O.conv2d(x, w, None, [1, 1], 1, pads, strides), RTOL, "thin conv2d %s" % (shape,))


def _depthwise_case():
    rng = np.random.default_rng(31)
    x = rng.standard_normal((2, 3, 6, 24)).astype(np.float32) + 1.5      # non-zero row starts: an over-read is visible
    w = rng.standard_normal((3, 1, 5, 5)).astype(np.float32)
    bias = rng.standard_normal(3).astype(np.float32)
    return x, w, bias


def test_x86_generic_depthwise_divergence_is_pinned():
    """DESIGN.md section 4 lists two deliberate divergences from lele's x86 build for depthwise convolutions outside the 3x3 / s1 /
    p1 case: upstream drops bias and SiLU there, and its 8-wide middle reads the next row's pixels at the right edge
    (conv2d.rs:535-570, 3130-3215).  oracle/npref.py restates that code; this test pins WHERE the restatement and the ONNX
    definition (what the device implements, and what the reference's own ORT comparisons expect) part ways."""
    from oracle import npref
    from oracle import pyoracle as O
    x, w, bias = _depthwise_case()
    got, defined = npref.depthwise_conv2d_x86_generic(x, w, (1, 1), (0, 0, 4, 4))
    plain = O.conv2d(x, w, None, (), 3, (0, 0, 4, 4), (1, 1))
    assert got.shape == plain.shape == (2, 3, 6, 24)                         # same output size
    # padding on the right only (an asymmetric "same" padding): the vector step for columns 16..23 reads columns up to 23 + 4,
    # i.e. its last four lanes see the first pixels of the next row where the definition has zeros
    edge = np.zeros(24, bool)
    edge[20:] = True
    assert np.allclose(got[..., ~edge], plain[..., ~edge], rtol=1e-5, atol=1e-5)        # away from the right edge: the definition
    assert not np.allclose(got[..., edge][defined[..., edge]], plain[..., edge][defined[..., edge]], rtol=1e-3, atol=1e-3)
    assert not defined[-1, -1, -1, 20:].all()                                # the very last row reads past the buffer upstream
    # bias and SiLU never reach that kernel upstream: the definition with them differs everywhere
    full = O.conv2d(x, w, bias, (), 3, (0, 0, 4, 4), (1, 1), act="silu")
    assert not np.allclose(full[..., ~edge], got[..., ~edge], rtol=1e-2, atol=1e-2)


@pytest.mark.gpu
def test_device_generic_depthwise_follows_the_definition_not_the_x86_edge(ctx):
    from lele_amd import kernels as K
    from oracle import npref
    from oracle import pyoracle as O
    x, w, bias = _depthwise_case()
    dev = K.conv2d_silu(x, w, bias, [1, 1], 3, [0, 0, 4, 4], [1, 1], ctx=ctx).numpy()
    want = O.conv2d(x, w, bias, (), 3, (0, 0, 4, 4), (1, 1), act="silu")
    assert np.allclose(dev, want, rtol=1e-4, atol=1e-5)
    x86, defined = npref.depthwise_conv2d_x86_generic(x, w, (1, 1), (0, 0, 4, 4))
    assert not np.allclose(dev[..., 20:][defined[..., 20:]], x86[..., 20:][defined[..., 20:]], rtol=1e-2, atol=1e-2)
