"""The oracle's SCALAR build mode (oracle/scalar.cpp; SURVEY.md section 7 step 1 / 8(c), BASELINE configs[0]): the restatement of the
`#[cfg(not(any(target_arch = "x86_64", target_arch = "aarch64", target_arch = "wasm32")))]` bodies of lele's kernels
(src/kernels/norm.rs:193-216, 286-306; rnn.rs:207-221; conv1d.rs:1578-1615; quantization.rs:1751-1796; the libm bodies of math.rs),
checked three ways: against float64 restatements of the SAME loops (known answers), against the reference's own scalar oracle for the
GRU (ref_gru_step, tests/regression_kernels.rs:602-737: the four cases and tolerances of the reference), and against the AVX2 mode --
the two modes are different roundings of the same mathematics and must agree to the f32 class, not bit for bit."""
import numpy as np
import pytest

from oracle import pyoracle as O
from tests.test_conv_rnn import GRU_KATS, _gru_tensors, _ref_gru


def _rel(a, b):
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def test_mode_is_scoped_and_off_by_default():
    assert not O.scalar_mode()
    with O.scalar():
        assert O.scalar_mode()
        with O.scalar():
            assert O.scalar_mode()
        assert O.scalar_mode()
    assert not O.scalar_mode()


def test_softmax_scalar_known_answer_and_against_avx2():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((5, 77)) * 4).astype(np.float32)
    with O.scalar():
        s = O.softmax(x, -1)
    # norm.rs:193-216 in float64: max, exp(x - max), sum in element order, one multiply by 1 / sum
    e = np.exp(x.astype(np.float64) - x.max(-1, keepdims=True))
    want = e / e.sum(-1, keepdims=True)
    assert _rel(s, want) < 5e-7 and np.allclose(s.sum(-1), 1.0, atol=1e-6)
    a = O.softmax(x, -1)
    assert _rel(s, a) < 1e-5 and not O.scalar_mode()
    # ORT constants of the reference's own test (tests/kernel_accuracy.rs:27-49 / verify_operators.rs:56-70)
    with O.scalar():
        y = O.softmax(np.array([[1.0, 2.0, 3.0]], np.float32), -1)
    assert np.abs(y - np.array([[0.09003057, 0.24472847, 0.66524096]], np.float32)).max() < 1e-6


def test_layer_norm_scalar_is_the_one_pass_variance_form():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((9, 130)) * 3 + 1.5).astype(np.float32)
    g, b = (1 + 0.1 * rng.standard_normal(130)).astype(np.float32), (0.1 * rng.standard_normal(130)).astype(np.float32)
    with O.scalar():
        y = O.layer_norm(x, g, b, -1, 1e-5)
    # norm.rs:286-306 step by step in f32: running sum and sum of squares in element order, var = E[x^2] - mean^2
    want = np.empty_like(x)
    inv_n = np.float32(1.0) / np.float32(130)
    for i, row in enumerate(x):
        s = q = np.float32(0)
        for v in row:
            s = np.float32(s + v)
            q = np.float32(q + np.float32(v * v))
        mean = np.float32(s * inv_n)
        var = np.float32(np.float32(q * inv_n) - np.float32(mean * mean))
        inv_std = np.float32(1.0) / np.sqrt(np.float32(var + np.float32(1e-5)), dtype=np.float32)
        want[i] = (((row - mean).astype(np.float32) * inv_std).astype(np.float32) * g).astype(np.float32) + b
    assert np.array_equal(y, want)
    assert _rel(y, O.layer_norm(x, g, b, -1, 1e-5)) < 1e-5


@pytest.mark.parametrize("kat", GRU_KATS, ids=[k[0] for k in GRU_KATS])
def test_gru_scalar_against_the_references_scalar_oracle(kat):
    """rnn.rs:319-349 mis-indexes the hidden gate's recurrent term (H + k for 2 H + k, rnn.rs:330); the scalar mode follows
    ref_gru_step, the reference's own scalar statement of the operator, at the reference's tolerances"""
    name, T, I, H, _, _, _, _, lbr, tol = kat
    x, w, r, b = _gru_tensors(kat)
    bw = b[0, :3 * H] if b is not None else np.zeros(3 * H)
    br = b[0, 3 * H:] if b is not None else np.zeros(3 * H)
    yref, href = _ref_gru(x[:, 0, :], w[0], r[0], bw, br, H)
    with O.scalar():
        y, h = O.gru(x, w, r, b)
    assert np.abs(y.reshape(T, H) - yref).max() <= tol and np.abs(h.ravel() - href).max() <= tol
    y2, h2 = O.gru(x, w, r, b)
    assert np.abs(y - y2).max() < 1e-5


def test_lstm_scalar_gate_stage():
    rng = np.random.default_rng(3)
    T, I, H = 6, 12, 20
    x = rng.standard_normal((T, 1, I)).astype(np.float32)
    w = (rng.standard_normal((1, 4 * H, I)) * 0.3).astype(np.float32)
    r = (rng.standard_normal((1, 4 * H, H)) * 0.3).astype(np.float32)
    b = (rng.standard_normal((1, 8 * H)) * 0.3).astype(np.float32)
    with O.scalar():
        y, h, c = O.lstm(x, w, r, b)
    # rnn.rs:152-154 + 207-221 in float64 (gate order i, o, f, c)
    hh, cc = np.zeros(H), np.zeros(H)
    sig = lambda v: 1.0 / (1.0 + np.exp(-v))
    for t in range(T):
        g = w[0].astype(np.float64) @ x[t, 0] + r[0].astype(np.float64) @ hh + b[0, :4 * H] + b[0, 4 * H:]
        i_g, o_g, f_g, c_g = sig(g[:H]), sig(g[H:2 * H]), sig(g[2 * H:3 * H]), np.tanh(g[3 * H:])
        cc = f_g * cc + i_g * c_g
        hh = o_g * np.tanh(cc)
    assert np.abs(h.ravel() - hh).max() < 1e-5 and np.abs(c.ravel() - cc).max() < 1e-5
    y2, h2, c2 = O.lstm(x, w, r, b)
    assert np.abs(h - h2).max() < 1e-5 and np.abs(y - y2).max() < 1e-5


def test_conv1d_single_channel_scalar_exact_answers():
    # the exact-integer cases of conv1d.rs:1621-1674 shape: one input channel, no padding; plus a long-kernel STFT-like case
    x = np.arange(1, 11, dtype=np.float32).reshape(1, 1, 10)
    w = np.array([[[1, 0, -1]], [[2, 1, 0]]], np.float32)
    bias = np.array([0.5, -30.0], np.float32)
    with O.scalar():
        y = O.conv1d(x, w, bias, [1], 1, [0, 0], [2], True)
    want = np.stack([[max(x[0, 0, t] - x[0, 0, t + 2] + 0.5, 0) for t in range(0, 8, 2)],
                     [max(2 * x[0, 0, t] + x[0, 0, t + 1] - 30.0, 0) for t in range(0, 8, 2)]])[None].astype(np.float32)
    assert np.array_equal(y, want)
    rng = np.random.default_rng(5)
    xs = rng.standard_normal((2, 1, 640)).astype(np.float32)
    ws = (rng.standard_normal((258, 1, 256)) / 16).astype(np.float32)
    with O.scalar():
        ys = O.conv1d(xs, ws, None, [1], 1, [0, 0], [128])
    ya = O.conv1d(xs, ws, None, [1], 1, [0, 0], [128])
    assert ys.shape == ya.shape == (2, 258, 4) and _rel(ys, ya) < 2e-6
    # the running sum in tap order, in f32
    t, oc = 3, 17
    s = np.float32(0)
    for k in range(256):
        s = np.float32(s + np.float32(xs[1, 0, 128 * t + k] * ws[oc, 0, k]))
    assert ys[1, oc, t] == s


def test_dynamic_quantize_linear_scalar_rounds_half_away():
    # quantization.rs:1751-1796: two roundings (x * inv_scale, + zp) and f32::round for EVERY element; the AVX2 body rounds
    # fma(x, inv_scale, zp) half to even -- codes differ exactly at the ties
    x = np.array([-1.0, 0.0, 0.5, 1.5, 2.5, 3.5, 254.0], np.float32)   # range 255: scale 1, zp 1
    with O.scalar():
        y, sc, zp = O.dynamic_quantize_linear(x)
    assert sc[0] == np.float32(1.0) and zp[0] == 1.0
    assert y.tolist() == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 255.0]          # 1.5 -> 2, 2.5 -> 3, 3.5 -> 4, 4.5 -> 5: half away
    ya, sa, za = O.dynamic_quantize_linear(np.concatenate([x, np.zeros(1, np.float32)]))   # 8 elements: all in the SIMD body
    assert sa[0] == sc[0] and za[0] == zp[0] and ya[:7].tolist() == [0.0, 1.0, 2.0, 2.0, 4.0, 4.0, 255.0]   # half to even
    rng = np.random.default_rng(9)
    xr = (rng.standard_normal(4001) * 3).astype(np.float32)
    with O.scalar():
        ys, ss, zs = O.dynamic_quantize_linear(xr)
    ya, sa, za = O.dynamic_quantize_linear(xr)
    assert ss[0] == sa[0] and zs[0] == za[0] and np.abs(ys - ya).max() <= 1.0 and (ys != ya).mean() < 1e-2


def test_unary_scalar_mode_is_libm():
    x = np.linspace(-6, 6, 41).astype(np.float32)
    with O.scalar():
        sg, th, ex, ge = O.unary("sigmoid", x), O.unary("tanh", x), O.unary("exp", x), O.unary("gelu", x)
    from math import erf
    assert _rel(sg, 1 / (1 + np.exp(-x.astype(np.float64)))) < 3e-7 and _rel(th, np.tanh(x.astype(np.float64))) < 3e-7
    assert _rel(ex, np.exp(x.astype(np.float64))) < 3e-7
    assert _rel(ge, np.array([v * 0.5 * (1 + erf(v * 0.7071067811865475)) for v in x.astype(np.float64)])) < 1e-6
    assert _rel(sg, O.unary("sigmoid", x)) < 1e-6   # the polynomial body is the same function to the f32 class
