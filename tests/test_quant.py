"""Dynamic-quantised u8 linear path: oracle pinned on the reference's KATs (CPU); device BIT-EXACT vs oracle (GPU)."""
import json
import os

import numpy as np
import pytest

K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def _arr(k, name):
    return np.array(k[name], np.float32).reshape(k[name + "_shape"])


def _sweep_inputs(m, k, n):
    a = ((np.arange(m * k, dtype=np.int64) * 7 + 13) % 256).astype(np.float32).reshape(m, k)
    b = ((np.arange(k * n, dtype=np.int64) * 11 + 3) % 256).astype(np.float32).reshape(k, n)
    return a, b


def test_oracle_kats(orc):
    for name in ("mat_mul_integer_zero_points", "mat_mul_integer_no_zp"):
        k = K[name]
        r = orc.mat_mul_integer(_arr(k, "a"), _arr(k, "b"), [k["zp_a"]], [k["zp_b"]])
        assert np.abs(r.ravel() - np.array(k["expected"])).max() < k["tol"]
    k = K["mat_mul_integer_per_channel_scale"]
    r = orc.mat_mul_integer(_arr(k, "a"), _arr(k, "b"), None, None, k["scale"])
    assert r.shape == (1, 2) and np.abs(r.ravel() - np.array(k["expected"])).max() < k["tol"]
    k = K["dynamic_quantize_accuracy"]
    x = _arr(k, "x")
    y, s, z = orc.dynamic_quantize_linear(x)
    assert np.abs((y - z[0]) * s[0] - x).max() < s[0] + 0.1  # kernel_accuracy.rs:239-244


def test_oracle_int8_shape_sweep(orc):
    k = K["int8_shape_sweep"]
    for m, kk, n in k["shapes"]:
        if m * kk * n > 3_000_000:  # keep the CPU suite fast; the big shapes run in the GPU test below
            continue
        a, b = _sweep_inputs(m, kk, n)
        ref = (a.astype(np.float64) - k["zp_a"]) @ (b.astype(np.float64) - k["zp_b"])  # ref_mat_mul_integer
        got = orc.mat_mul_integer(a, b, [k["zp_a"]], [k["zp_b"]])
        assert np.array_equal(got, ref.astype(np.float32)), (m, kk, n)


def test_oracle_fused_matches_unfused_composition(orc):
    # fused == dynamic_quantize_linear + mat_mul_integer(scale = dyn*w_scale, bias) when the batch is one slice and
    # K is a multiple of 8 (no scalar-tail rounding difference), cf. SURVEY.md section 7 defect (v)/(vi)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((1, 9, 32)).astype(np.float32)
    w = np.clip(np.round(128 + 32 * rng.standard_normal((32, 24))), 0, 255).astype(np.float32)
    ws = (np.abs(rng.standard_normal(24)) * 0.01 + 0.002).astype(np.float32)
    bias = (rng.standard_normal(24) * 0.02).astype(np.float32)
    fused = orc.fused_quantized_linear(x, w, ws, [128.0], bias, relu=True)
    y, s, z = orc.dynamic_quantize_linear(x)
    comp = orc.mat_mul_integer(y, w, z, [128.0], (s[0] * ws).astype(np.float32), bias, relu=True)
    assert np.array_equal(fused, comp)


def _weights(rng, k, n):
    w = np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)
    ws = (np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)
    bias = (rng.standard_normal(n) * 0.02).astype(np.float32)
    return w, ws, bias


@pytest.mark.gpu
def test_device_kats(ctx):
    from lele_amd import kernels as Kk
    for name in ("mat_mul_integer_zero_points", "mat_mul_integer_no_zp"):
        k = K[name]
        r = Kk.mat_mul_integer(_arr(k, "a"), _arr(k, "b"), [k["zp_a"]], [k["zp_b"]], ctx=ctx)
        assert r.shape == (2, 2) and np.abs(r.data - np.array(k["expected"])).max() < k["tol"]
    k = K["mat_mul_integer_per_channel_scale"]
    r = Kk.mat_mul_integer_with_scale_bias(_arr(k, "a"), _arr(k, "b"), None, None, np.array(k["scale"], np.float32),
                                           None, ctx=ctx)
    assert np.abs(r.data - np.array(k["expected"])).max() < k["tol"]


@pytest.mark.gpu
def test_device_int8_shape_sweep_exact(ctx):
    from lele_amd import kernels as Kk
    k = K["int8_shape_sweep"]
    for m, kk, n in k["shapes"]:
        a, b = _sweep_inputs(m, kk, n)
        ref = ((a.astype(np.float64) - k["zp_a"]) @ (b.astype(np.float64) - k["zp_b"])).astype(np.float32)
        got = Kk.mat_mul_integer(a, b, [k["zp_a"]], [k["zp_b"]], ctx=ctx)
        assert got.shape == (m, n) and np.array_equal(got.numpy(), ref), (m, kk, n)


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 93, 560, 1536), (1, 93, 512, 512), (1, 93, 512, 2048), (1, 93, 2048, 512),
                                   (3, 17, 64, 40), (2, 5, 37, 19), (1, 1, 8, 1), (1, 504, 512, 1536), (4, 171, 512, 512),
                                   (1, 33, 100, 130)])
@pytest.mark.parametrize("relu", [False, True])
def test_device_fused_quantized_linear_bit_exact(ctx, orc, shape, relu):
    from lele_amd import kernels as Kk
    b, m, k, n = shape
    rng = np.random.default_rng(b * 1000 + m + k + n)
    x = (rng.standard_normal((b, m, k)) * rng.uniform(0.5, 3.0, (b, 1, 1))).astype(np.float32)  # a range per slice
    w, ws, bias = _weights(rng, k, n)
    ref = orc.fused_quantized_linear(x, w, ws, [128.0], bias, relu)
    got = Kk.fused_quantized_linear(x, w, ws, np.array([128.0], np.float32), bias, relu, ctx=ctx)
    assert got.shape == ref.shape and np.array_equal(got.numpy(), ref)


@pytest.mark.gpu
def test_device_fused_variants(ctx, orc):
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(9)
    x = rng.standard_normal((2, 7, 48)).astype(np.float32)
    w, ws, bias = _weights(rng, 48, 20)
    # scalar weight scale, empty bias, non-default weight zero point, weights declared immutable (cached pre-pack)
    for wsc, bb, wz in ((ws[:1], bias, 128.0), (ws, np.zeros((0,), np.float32), 131.0), (ws, None, 0.0)):
        ref = orc.fused_quantized_linear(x, w, wsc, [wz], bb)
        wq = Kk.Weight(w)
        for _ in range(2):  # second call hits the weight cache
            got = Kk.fused_quantized_linear(x, wq, wsc, np.array([wz], np.float32), bb, ctx=ctx)
            assert np.array_equal(got.numpy(), ref)
    # all-zero and constant inputs (range clamps to 1e-5)
    for xx in (np.zeros((1, 4, 48), np.float32), np.full((1, 4, 48), 0.25, np.float32)):
        assert np.array_equal(Kk.fused_quantized_linear(xx, w, ws, [128.0], bias, ctx=ctx).numpy(),
                              orc.fused_quantized_linear(xx, w, ws, [128.0], bias))


@pytest.mark.gpu
def test_device_dynamic_quantize_and_unfused_chain_bit_exact(ctx, orc):
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(4)
    for shape in ((2, 4), (3, 5, 7), (1, 93, 512), (13,)):
        x = (rng.standard_normal(shape) * 2 + 0.3).astype(np.float32)
        y, s, z = Kk.dynamic_quantize_linear(x, ctx=ctx)
        ry, rs, rz = orc.dynamic_quantize_linear(x)
        assert y.shape == x.shape and np.array_equal(y.numpy(), ry)
        assert np.array_equal(s.numpy(), rs) and np.array_equal(z.numpy(), rz)
    x = rng.standard_normal((2, 9, 40)).astype(np.float32)
    w, ws, bias = _weights(rng, 40, 24)
    y, s, z = Kk.dynamic_quantize_linear(x, ctx=ctx)  # device-resident chain: no host round trip for y / z
    comb = (orc.dynamic_quantize_linear(x)[1][0] * ws).astype(np.float32)
    got = Kk.mat_mul_integer_with_scale_bias_relu(y, w, z, np.array([128.0], np.float32), comb, bias, ctx=ctx)
    ry, rs, rz = orc.dynamic_quantize_linear(x)
    assert np.array_equal(got.numpy(), orc.mat_mul_integer(ry, w, rz, [128.0], comb, bias, relu=True))
    # batched B (batch_b == batch_a) and broadcast A
    a = rng.integers(0, 256, (3, 6, 20)).astype(np.float32)
    bq = rng.integers(0, 256, (3, 20, 10)).astype(np.float32)
    assert np.array_equal(Kk.mat_mul_integer(a, bq, [3.0], [250.0], ctx=ctx).numpy(),
                          orc.mat_mul_integer(a, bq, [3.0], [250.0]))
    assert np.array_equal(Kk.mat_mul_integer(a[:1], bq, [3.0], [250.0], ctx=ctx).numpy(),
                          orc.mat_mul_integer(a[:1], bq, [3.0], [250.0]))


@pytest.mark.gpu
def test_layer_norm_row_statistics_feed_the_dynamic_quantisation(ctx):
    """LayerNorm leaves per-row {min, max} next to its result; a fused_quantized_linear reading that buffer skips its own
    range pass (quant.hip).  The result must not change by a bit -- against the oracle and against the same call on a copy
    of the activation (which has no statistics attached) -- and a buffer that was rewritten must not reuse stale ones."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    from oracle import pyoracle as O
    rng = np.random.default_rng(77)
    for b, m, k, n in ((1, 504, 512, 384), (32, 171, 512, 256), (3, 7, 560, 64), (2, 5, 1000, 40)):
        x = (rng.standard_normal((b, m, k)) * 2).astype(np.float32)
        g, be = (1 + 0.1 * rng.standard_normal(k)).astype(np.float32), (0.1 * rng.standard_normal(k)).astype(np.float32)
        w = Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
        ws, wz, bias = Weight((rng.random(n) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)), Weight(rng.standard_normal(n).astype(np.float32))
        buf = ctx.buf()
        xn = K.layer_norm(x, g, be, -1, 1e-5, out=buf, ctx=ctx)
        got = K.fused_quantized_linear(xn, w, ws, wz, bias, False, ctx=ctx).numpy()
        xn_host = xn.numpy()
        assert np.array_equal(xn_host, O.layer_norm(x, g, be, -1, 1e-5))
        want = O.fused_quantized_linear(xn_host, w.arr, ws.arr, wz.arr, bias.arr, False)
        assert np.array_equal(got, want), (b, m, k, n)
        assert np.array_equal(K.fused_quantized_linear(xn_host, w, ws, wz, bias, False, ctx=ctx).numpy(), want)   # no statistics: own range pass
        # rewrite the same buffer with different data of the same shape: the old statistics must not be used
        y = K.mul(xn, np.array([3.0], np.float32), out=buf, ctx=ctx)
        assert np.array_equal(K.fused_quantized_linear(y, w, ws, wz, bias, False, ctx=ctx).numpy(),
                              O.fused_quantized_linear(xn_host * np.float32(3.0), w.arr, ws.arr, wz.arr, bias.arr, False)), (b, m, k, n)


@pytest.mark.gpu
def test_residual_adds_in_the_linear_epilogue_are_bit_identical(ctx):
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(5)
    for b, m, k, n, relu in ((1, 504, 512, 512, False), (32, 171, 512, 512, False), (2, 33, 2048, 512, True), (1, 7, 64, 40, False), (3, 200, 512, 1536, False)):
        x = rng.standard_normal((b, m, k)).astype(np.float32)
        w = Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
        ws, wz, bias = Weight((rng.random(n) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)), Weight(rng.standard_normal(n).astype(np.float32))
        r1, r2 = rng.standard_normal((b, m, n)).astype(np.float32), rng.standard_normal((b, m, n)).astype(np.float32)
        lin = K.fused_quantized_linear(x, w, ws, wz, bias, relu, ctx=ctx)
        want1 = K.add(lin, r1, ctx=ctx).numpy()
        want2 = K.add3(lin, r1, r2, ctx=ctx).numpy()
        assert np.array_equal(K.fused_quantized_linear_residual(x, w, ws, wz, bias, relu, r1, ctx=ctx).numpy(), want1), (b, m, k, n)
        assert np.array_equal(K.fused_quantized_linear_residual(x, w, ws, wz, bias, relu, r1, r2, ctx=ctx).numpy(), want2), (b, m, k, n)
        # a broadcasting residual takes the two-pass route
        assert np.array_equal(K.fused_quantized_linear_residual(x, w, ws, wz, bias, relu, r1[:, :1], ctx=ctx).numpy(), K.add(lin, r1[:, :1], ctx=ctx).numpy())


@pytest.mark.gpu
def test_fused_ffn_never_stores_the_hidden_layer_and_keeps_every_bit(ctx):
    """lele_hip_fused_ffn_quantized = two fused_quantized_linear calls with a ReLU between them.  Large hidden layers take the
    recompute route (the first product twice: range, then quantise to i8 + row sums); the result must equal the oracle's
    composition and the two separate device calls bit for bit -- slices that straddle row tiles, K not a multiple of the tile,
    odd output widths, with and without residuals, and the shapes that fall back to the two calls."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    from oracle import pyoracle as O
    rng = np.random.default_rng(11)

    def lin(k, n):
        return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)),
                Weight((rng.random(n) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
                Weight(rng.standard_normal(n).astype(np.float32) * 0.1))
    for b, m, k1, n1, n2, relu2 in ((32, 171, 512, 2048, 512, False), (16, 300, 200, 2048, 72, True), (1, 4800, 512, 2048, 512, False), (9, 1000, 500, 2048, 260, False),
                                    (2, 33, 64, 128, 32, False), (1, 504, 512, 2048, 512, False), (40, 129, 96, 2176, 64, False),
                                    (70, 128, 500, 1024, 72, False), (5, 2001, 512, 640, 36, True)):
        x = (rng.standard_normal((b, m, k1)) * rng.uniform(0.3, 3.0, (b, 1, 1))).astype(np.float32)
        w1, w2 = lin(k1, n1), lin(n1, n2)
        r1, r2 = rng.standard_normal((b, m, n2)).astype(np.float32), rng.standard_normal((b, m, n2)).astype(np.float32)
        hid = O.fused_quantized_linear(x, w1[0].arr, w1[1].arr, w1[2].arr, w1[3].arr, True)
        want = O.fused_quantized_linear(hid, w2[0].arr, w2[1].arr, w2[2].arr, w2[3].arr, relu2)
        got = K.fused_ffn_quantized(x, *w1, *w2, relu2, ctx=ctx).numpy()
        assert got.shape == want.shape and np.array_equal(got, want), (b, m, k1, n1, n2)
        # the two calls on the device, and the residual forms
        h_dev = K.fused_quantized_linear(x, *w1, True, ctx=ctx)
        assert np.array_equal(h_dev.numpy(), hid)
        two = K.fused_quantized_linear_residual(h_dev, *w2, relu2, r1, r2, ctx=ctx).numpy()
        assert np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, relu2, r1, r2, ctx=ctx).numpy(), two), (b, m, k1, n1, n2)
        one = K.fused_quantized_linear_residual(h_dev, *w2, relu2, r1, ctx=ctx).numpy()
        assert np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, relu2, r1, ctx=ctx).numpy(), one), (b, m, k1, n1, n2)
        with _env(LELE_HIP_FFN_FUSED=0):
            assert np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, relu2, r1, r2, ctx=ctx).numpy(), two)
    # input statistics left by a LayerNorm (the model's case) are used as before, and a slice of all-negative pre-activations
    # (hidden layer identically zero: range floor 1e-5) quantises like the stored tensor would
    x = rng.standard_normal((32, 171, 512)).astype(np.float32)
    g, be = np.ones(512, np.float32), np.zeros(512, np.float32)
    xn = K.layer_norm(x, g, be, -1, 1e-5, out=ctx.buf(), ctx=ctx)
    w1, w2 = lin(512, 2048), lin(2048, 512)
    w1 = (w1[0], w1[1], w1[2], Weight(np.full(2048, -1e4, np.float32)))     # bias drives every pre-activation below zero
    hid = O.fused_quantized_linear(xn.numpy(), w1[0].arr, w1[1].arr, w1[2].arr, w1[3].arr, True)
    assert not hid.any()
    want = O.fused_quantized_linear(hid, w2[0].arr, w2[1].arr, w2[2].arr, w2[3].arr, False)
    assert np.array_equal(K.fused_ffn_quantized(xn, *w1, *w2, False, ctx=ctx).numpy(), want)
    # weight scales of either sign, and zeros of either sign (the range pass of the tiled route evaluates its epilogue only at the
    # extreme i32 totals of a column: the maximum for a scale >= 0, the minimum below)
    for b, m in ((32, 171), (1, 4800), (9, 1000)):
        x = (rng.standard_normal((b, m, 512)) * rng.uniform(0.3, 3.0, (b, 1, 1))).astype(np.float32)
        w1, w2 = lin(512, 2048), lin(2048, 512)
        sc = w1[1].arr.copy()
        sc[::3] *= -1
        sc[5::97] = 0.0
        sc[7::101] = -0.0
        w1 = (w1[0], Weight(sc), w1[2], w1[3])
        hid = O.fused_quantized_linear(x, w1[0].arr, sc, w1[2].arr, w1[3].arr, True)
        want = O.fused_quantized_linear(hid, w2[0].arr, w2[1].arr, w2[2].arr, w2[3].arr, False)
        for rs in (1, 0):
            with _env(LELE_HIP_IGEMM_RS=rs):
                assert np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, False, ctx=ctx).numpy(), want), (b, m, rs)


@pytest.mark.gpu
@pytest.mark.parametrize("b,m,k,n,relu,bias,scalar_ws", [
    (1, 8300, 512, 1028, False, True, False),    # one slice, rows not a multiple of 32, a partial last column tile (n % 32 = 4)
    (9, 1000, 500, 1536, True, True, False),     # K padded to 512 with zero bytes; slices straddle row tiles at every offset
    (70, 128, 512, 1024, False, False, False),   # slices of exactly four row tiles; no bias; the narrowest result the route takes
    (3, 3333, 512, 1536, False, True, True),     # one weight scale for all columns; 9999 rows
    (64, 171, 512, 1156, True, True, False),     # the configs[3] row structure with a ragged result
    (64, 171, 512, 196, True, True, False),      # narrow results and stand-alone K = 2048 stay on the tiled kernels (measured
    (32, 171, 2048, 512, True, True, False),     # slower on this route): same bits either way
])
def test_register_stationary_gemm_bit_exact_on_edge_shapes(ctx, b, m, k, n, relu, bias, scalar_ws):
    """igemm_rs_kernel serves the quantised linears with K padded to 512 bytes, results at least 1024 wide and enough 32 x 32
    tiles to fill the chip (the K-split igemm_rs_ks4_kernel runs inside the fused feed-forward block: test_fused_ffn_*):
    bit-exact with the oracle on shapes chosen for their edges -- row tiles that end inside a slice,
    column tiles that end inside a 4-column group, slices of every alignment, rows % 32 != 0 -- and identical to the tiled
    kernels' result (LELE_HIP_IGEMM_RS=0)"""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    from oracle import pyoracle as O
    rng = np.random.default_rng(b * 7 + m + n)
    x = (rng.standard_normal((b, m, k)) * rng.uniform(0.2, 4.0, (b, 1, 1))).astype(np.float32)
    w = Weight(np.clip(np.round(128 + 40 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
    ws = Weight((np.full(1, 0.0071) if scalar_ws else rng.random(n) * 0.01 + 0.002).astype(np.float32))
    wz = Weight(np.array([121.0], np.float32))          # a weight zero point other than 128: the row-sum term is live
    bs = Weight(rng.standard_normal(n).astype(np.float32)) if bias else None
    want = O.fused_quantized_linear(x, w.arr, ws.arr, wz.arr, bs.arr if bias else None, relu)
    got = K.fused_quantized_linear(x, w, ws, wz, bs, relu, ctx=ctx).numpy()
    assert got.shape == want.shape and np.array_equal(got, want)
    with _env(LELE_HIP_IGEMM_RS=0):
        assert np.array_equal(K.fused_quantized_linear(x, w, ws, wz, bs, relu, ctx=ctx).numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("b,m,n,relu,bias,scalar_ws", [
    (1, 5472, 512, False, True, False),     # the projection behind the attention on a configs[3] shard, as one slice
    (32, 171, 512, False, True, False),     # ... and with its 32 slices (a row tile holds rows of two of them)
    (150, 7, 100, True, True, False),       # slices shorter than a wave's 16 rows: up to four parameter sets a wave; n % 32 = 4
    (3, 347, 36, False, False, True),       # 1041 rows (a last tile of 17), two column tiles of which one is partial, no bias
    (33, 64, 260, True, True, False),       # slices of exactly two row tiles; nine column tiles: the last wave has one
    (1, 1025, 4, False, True, True),        # the narrowest result; one row in the last tile
    (1, 504, 512, False, True, False),      # one 30 s utterance: 16 row tiles -> four column tiles a workgroup, one a wave, grid 16 x 4
    (2, 300, 384, True, True, False),       # 19 row tiles, 12 column tiles in three groups of four; slices end inside a tile
    (1, 1600, 500, False, False, True),     # 50 row tiles -> eight column tiles a workgroup; the second group has a partial last tile
])
def test_activation_stationary_gemm_bit_exact_on_edge_shapes(ctx, b, m, n, relu, bias, scalar_ws):
    """igemm_as_kernel (K = 512 exactly, results at most 512 wide, 256 rows or more): rows quantised inside the GEMM with the
    slices' parameters reduced from the {min, max} partials on the spot -- same bits as the oracle and as the tiled route
    (LELE_HIP_IGEMM_RS=0), with and without the residual operands of the epilogue, and the parameters it publishes / the block
    statistics it leaves serve a following quantised linear"""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    from oracle import pyoracle as O
    rng = np.random.default_rng(b * 11 + m + n)
    k = 512
    x = (rng.standard_normal((b, m, k)) * rng.uniform(0.2, 4.0, (b, 1, 1))).astype(np.float32)
    w = Weight(np.clip(np.round(128 + 40 * rng.standard_normal((k, n))), 0, 255).astype(np.float32))
    ws = Weight((np.full(1, 0.0071) if scalar_ws else rng.random(n) * 0.01 + 0.002).astype(np.float32))
    wz = Weight(np.array([121.0], np.float32))
    bs = Weight(rng.standard_normal(n).astype(np.float32)) if bias else None
    want = O.fused_quantized_linear(x, w.arr, ws.arr, wz.arr, bs.arr if bias else None, relu)
    lin = K.fused_quantized_linear(x, w, ws, wz, bs, relu, ctx=ctx)
    assert lin.shape == want.shape and np.array_equal(lin.numpy(), want)
    r1, r2 = rng.standard_normal(want.shape).astype(np.float32), rng.standard_normal(want.shape).astype(np.float32)
    assert np.array_equal(K.fused_quantized_linear_residual(x, w, ws, wz, bs, relu, r1, ctx=ctx).numpy(), want + r1)
    assert np.array_equal(K.fused_quantized_linear_residual(x, w, ws, wz, bs, relu, r1, r2, ctx=ctx).numpy(), (want + r1) + r2)
    with _env(LELE_HIP_IGEMM_RS=0):
        assert np.array_equal(K.fused_quantized_linear(x, w, ws, wz, bs, relu, ctx=ctx).numpy(), want)
    if n % 4 == 0 and n >= 16:  # the result feeds another quantised linear: its range comes from what this kernel left (or a scan)
        w2 = _qw(rng, n, 24)
        assert np.array_equal(K.fused_quantized_linear(lin, *w2, False, ctx=ctx).numpy(),
                              O.fused_quantized_linear(want, w2[0].arr, w2[1].arr, [128.0], w2[3].arr, False))


@pytest.mark.gpu
def test_fused_ffn_is_deterministic_under_repetition(ctx):
    """the recompute route combines per-workgroup maxima through LDS and global atomics and row sums through integer atomics:
    300 repetitions of one configs[3]-shard call must give one bit pattern (a race here shows up as one utterance in a few
    hundred calls quantised with a neighbour's range), equal to the tiled-kernel route's"""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(0)

    def lin(k, n):
        return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)),
                Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
                Weight((rng.standard_normal(n) * 0.02).astype(np.float32)))
    w1, w2 = lin(512, 2048), lin(2048, 512)
    x = ctx.buf().upload((rng.standard_normal((32, 171, 512)) * rng.uniform(0.3, 3, (32, 1, 1))).astype(np.float32))
    with _env(LELE_HIP_IGEMM_RS=0):
        ref = K.fused_ffn_quantized(x, *w1, *w2, False, ctx=ctx).numpy().copy()
    ob = ctx.buf()
    bad = [it for it in range(300) if not np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx).numpy(), ref)]
    assert not bad, "iterations with different bits: %s" % bad[:10]


@pytest.mark.gpu
@pytest.mark.parametrize("b,m", [(32, 171), (1, 504), (1, 4800), (7, 700), (64, 80), (3, 33 * 32)])
def test_fused_ffn_one_launch_form_is_bit_identical_to_the_two_launches(ctx, b, m):
    """igemm_rs_kernel<3>: the range pass and the quantise pass of the feed-forward block's first product in ONE launch -- every row tile
    stays in LDS, a workgroup publishes its maxima, waits for the row ranges that share a slice with its own and runs the products again.
    LELE_HIP_FFN_ONE_LAUNCH=2 makes the call fail rather than fall back, =0 keeps the two launches: same bits, equal to the oracle's two
    calls; slices that straddle row tiles and row ranges (171, 700, 80 rows), one slice over the whole grid (504, 4800), slices of whole
    tiles; eagerly, under repetition (a workgroup that read a neighbour's maximum too early shows up as one utterance in a few hundred
    calls) and as replays of a recorded graph (the counters are never cleared: every launch continues where the last one stopped)."""
    import gc
    from lele_amd import kernels as K
    from oracle import pyoracle as O
    gc.collect()          # contexts other tests dropped without closing them are destroyed now
    rng = np.random.default_rng(b * 13 + m)
    w1, w2 = _qw(rng, 512, 2048), _qw(rng, 2048, 512)
    xh = (rng.standard_normal((b, m, 512)) * rng.uniform(0.3, 3.0, (b, 1, 1))).astype(np.float32)
    x = ctx.buf().upload(xh)
    hid = O.fused_quantized_linear(xh, w1[0].arr, w1[1].arr, w1[2].arr, w1[3].arr, True)
    want = O.fused_quantized_linear(hid, w2[0].arr, w2[1].arr, w2[2].arr, w2[3].arr, False)
    ob = ctx.buf()
    with _env(LELE_HIP_FFN_ONE_LAUNCH=0):
        two = K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx).numpy().copy()
    assert np.array_equal(two, want)
    with _env(LELE_HIP_FFN_ONE_LAUNCH=2):
        one = K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx).numpy().copy()
        assert np.array_equal(one, want), (b, m)
        bad = [it for it in range(100) if not np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx).numpy(), want)]
        assert not bad, "iterations with different bits: %s" % bad[:10]
        # another input through the same counters, then a recorded graph of six calls replayed (launch numbers keep counting)
        x2h = (xh * np.float32(0.37) + np.float32(0.01)).astype(np.float32)
        x2 = ctx.buf().upload(x2h)
        hid2 = O.fused_quantized_linear(x2h, w1[0].arr, w1[1].arr, w1[2].arr, w1[3].arr, True)
        want2 = O.fused_quantized_linear(hid2, w2[0].arr, w2[1].arr, w2[2].arr, w2[3].arr, False)
        ob2 = ctx.buf()
        assert np.array_equal(K.fused_ffn_quantized(x2, *w1, *w2, False, out=ob2, ctx=ctx).numpy(), want2)
        ctx.sync()
        ctx.graph_begin()
        for _ in range(3):
            r1 = K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx)
            r2 = K.fused_ffn_quantized(x2, *w1, *w2, False, out=ob2, ctx=ctx)
        g = ctx.graph_end()
        for _ in range(20):
            g.launch()
        assert np.array_equal(r1.numpy(), want) and np.array_equal(r2.numpy(), want2)
        g.close()


@pytest.mark.gpu
def test_fused_ffn_one_launch_form_is_not_taken_beside_other_work(ctx):
    """the one-launch form waits for other workgroups, so it must be alone on the device: with a side lane in flight, on a side lane, or
    with a second context alive the call takes the two launches (LELE_HIP_FFN_ONE_LAUNCH=2 then fails by name) -- and gives the same bits"""
    import gc
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import LeleError
    gc.collect()
    rng = np.random.default_rng(3)
    w1, w2 = _qw(rng, 512, 2048), _qw(rng, 2048, 512)
    x = ctx.buf().upload(rng.standard_normal((32, 171, 512)).astype(np.float32))
    ob = ctx.buf()
    with _env(LELE_HIP_FFN_ONE_LAUNCH=2):
        want = K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx).numpy().copy()    # (a buffer that grows drains every lane)
        ctx.sync()
        ctx.lane_set(1)
        try:
            with pytest.raises(LeleError, match="one-launch"):
                K.fused_ffn_quantized(x, *w1, *w2, False, ctx=ctx)
        finally:
            ctx.lane_set(0)
        with pytest.raises(LeleError, match="one-launch"):     # lane 1 has been used since the streams were last drained
            K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx)
        ctx.sync()
        assert np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, False, out=ob, ctx=ctx).numpy(), want)
        other = lele_amd._lib.Ctx(0)
        try:
            with pytest.raises(LeleError, match="one-launch"):
                K.fused_ffn_quantized(x, *w1, *w2, False, ctx=ctx)
            with _env(LELE_HIP_FFN_ONE_LAUNCH=1):
                assert np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, False, ctx=ctx).numpy(), want)
        finally:
            other.close()
        assert np.array_equal(K.fused_ffn_quantized(x, *w1, *w2, False, ctx=ctx).numpy(), want)


@pytest.mark.gpu
def test_gemm_block_statistics_feed_the_next_dynamic_quantisation(ctx):
    """The small-problem GEMM kernels publish one {min, max} pair per workgroup; a single-slice quantised linear that reads
    the result next uses them instead of scanning it.  Same bits as the scan (compared with the call on a host copy)."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(91)

    def qw(k, n):
        return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)),
                Weight((rng.random(n) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
                Weight(rng.standard_normal(n).astype(np.float32)))
    for t in (504, 93, 7):
        # ffn1 (ReLU, small-problem i8 kernel) -> ffn2
        x = rng.standard_normal((1, t, 512)).astype(np.float32)
        w1, w2 = qw(512, 2048), qw(2048, 512)
        h = K.fused_quantized_linear(x, *w1, True, ctx=ctx)
        got = K.fused_quantized_linear(h, *w2, False, ctx=ctx).numpy()
        assert np.array_equal(got, K.fused_quantized_linear(h.numpy(), *w2, False, ctx=ctx).numpy()), t
        # P.V stored transposed (small-problem f32 kernel) -> output projection
        pr = rng.random((1, 4, t, t)).astype(np.float32)
        qkv = rng.standard_normal((1, t, 1536)).astype(np.float32)
        cv = [["slice", 2, 1024, 512], ["reshape", [0, 0, 4, 128]], ["transpose", [0, 2, 1, 3]]]
        av = K.matmul_view(pr, [], qkv, cv, out_perm=[0, 2, 1, 3], out_reshape=[0, 0, 512], ctx=ctx)
        wo = qw(512, 512)
        got = K.fused_quantized_linear(av, *wo, False, ctx=ctx).numpy()
        assert np.array_equal(got, K.fused_quantized_linear(av.numpy(), *wo, False, ctx=ctx).numpy()), t
    # more than one slice: the pairs do not apply (a tile may straddle two slices) -- the result must still be right
    x = rng.standard_normal((3, 40, 512)).astype(np.float32)
    h = K.fused_quantized_linear(x, *w1, True, ctx=ctx)
    assert np.array_equal(K.fused_quantized_linear(h, *w2, False, ctx=ctx).numpy(), K.fused_quantized_linear(h.numpy(), *w2, False, ctx=ctx).numpy())


def _qw(rng, k, n):
    from lele_amd._lib import Weight
    return (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)),
            Weight((np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
            Weight((rng.standard_normal(n) * 0.02).astype(np.float32)))


class _env:
    """set environment variables for the duration of a block (the library reads its tuning switches per call)"""

    def __init__(self, **kv):
        self.kv = {k: str(v) for k, v in kv.items()}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 93, 560, 1536), (1, 504, 512, 512), (1, 504, 512, 2048), (1, 504, 2048, 512), (32, 171, 512, 2048),
                                   (32, 171, 2048, 512), (4, 171, 560, 1536), (3, 17, 64, 40), (2, 40, 37, 19), (1, 1, 8, 1), (1, 33, 100, 130),
                                   (5, 32, 96, 64), (2, 64, 128, 33), (1, 504, 512, 25055), (3, 50, 256, 70), (2, 33, 300, 64),
                                   (2, 70, 1001, 96), (1, 9, 4096, 32)])
@pytest.mark.parametrize("relu", [False, True])
def test_device_quantized_linear_routes_bit_exact(ctx, orc, shape, relu):
    """declared-immutable weights, a device-resident activation: whatever route the sizes select (register-stationary kernels,
    tiled kernels, the small-problem kernel) is bit-exact against the oracle, and so is the tiled chain alone (LELE_HIP_IGEMM_RS=0)"""
    from lele_amd import kernels as Kk
    b, m, k, n = shape
    if relu and n > 4096:
        pytest.skip("the vocabulary-sized case runs once")
    rng = np.random.default_rng(b * 1000 + m + k + n)
    x = (rng.standard_normal((b, m, k)) * rng.uniform(0.5, 3.0, (b, 1, 1))).astype(np.float32)  # a range per slice
    w = _qw(rng, k, n)
    ref = orc.fused_quantized_linear(x, w[0].arr, w[1].arr, [128.0], w[3].arr, relu)
    xd = ctx.buf().upload(x)
    got = Kk.fused_quantized_linear(xd, *w, relu, ctx=ctx)
    assert got.shape == ref.shape and np.array_equal(got.numpy(), ref)
    with _env(LELE_HIP_IGEMM_RS=0):  # range pass | row quantisation | tiled i8 GEMM
        assert np.array_equal(Kk.fused_quantized_linear(xd, *w, relu, ctx=ctx).numpy(), ref)


@pytest.mark.gpu
@pytest.mark.parametrize("b,m,k,n,relu", [
    (1, 2125, 1024, 3844, False),    # 9 x 16 results of 256 x 256 with ragged last rows AND columns; K = 8 steps of 128 bytes
    (5, 700, 1152, 2048, True),      # slices that straddle the 256-row results; an odd number of K steps
    (2, 2048, 2048, 2052, False),    # one column block of four columns only
])
def test_compute_bound_i8_gemm_bit_exact(ctx, orc, b, m, k, n, relu):
    """igemm_big_kernel (csrc/igemm_big.h: 256 x 256 results, both operands by direct-to-LDS loads with the 16-byte chunks permuted on
    the global side) takes the i8 products with K a multiple of 128 >= 1024 and at least half a chip of results: the fused quantised
    linear (dynamic range per slice, scale + bias + ReLU, residual operands) and mat_mul_integer on it are bit-exact against the oracle"""
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(b * 1000 + m + k + n)
    x = (rng.standard_normal((b, m, k)) * rng.uniform(0.5, 3.0, (b, 1, 1))).astype(np.float32)
    w = _qw(rng, k, n)
    ref = orc.fused_quantized_linear(x, w[0].arr, w[1].arr, [128.0], w[3].arr, relu)
    xd = ctx.buf().upload(x)
    got = Kk.fused_quantized_linear(xd, *w, relu, ctx=ctx)
    assert got.shape == ref.shape and np.array_equal(got.numpy(), ref)
    r1, r2 = rng.standard_normal(ref.shape).astype(np.float32), rng.standard_normal(ref.shape).astype(np.float32)
    assert np.array_equal(Kk.fused_quantized_linear_residual(xd, *w, relu, r1, r2, ctx=ctx).numpy(), (ref + r1) + r2)
    a = rng.integers(0, 256, (b * m, k)).astype(np.float32)
    want = ((a.astype(np.float64) - 131.0) @ (w[0].arr.astype(np.float64) - 127.0)).astype(np.float32)   # exact: |total| < 2^31
    assert np.array_equal(Kk.mat_mul_integer(a, w[0], [131.0], [127.0], ctx=ctx).numpy(), want)


@pytest.mark.gpu
def test_gemm_statistics_feed_the_next_quantised_linear(ctx, orc):
    """a quantised linear leaves {min, max} pairs next to its result where its kernel has them; the quantised linear that reads
    the result next (ffn1 -> ffn2) derives its per-slice range from them instead of scanning the tensor.  Same bits as the oracle
    on the same input, for slices that straddle row tiles, on either route, and after the buffer was rewritten."""
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(123)
    for b, m, k, h, n in ((32, 171, 512, 2048, 512), (1, 504, 512, 2048, 512), (3, 40, 256, 288, 40), (2, 32, 256, 256, 64), (5, 100, 512, 270, 33),
                          (4, 31, 256, 256, 32), (3, 40, 64, 96, 40), (1, 5472, 512, 512, 512)):
        x = (rng.standard_normal((b, m, k)) * rng.uniform(0.5, 3.0, (b, 1, 1))).astype(np.float32)
        w1, w2 = _qw(rng, k, h), _qw(rng, h, n)
        hid_ref = orc.fused_quantized_linear(x, w1[0].arr, w1[1].arr, [128.0], w1[3].arr, True)
        out_ref = orc.fused_quantized_linear(hid_ref, w2[0].arr, w2[1].arr, [128.0], w2[3].arr, False)
        for rs in (1, 0):
            with _env(LELE_HIP_IGEMM_RS=rs):
                hbuf = ctx.buf()
                hid = Kk.fused_quantized_linear(ctx.buf().upload(x), *w1, True, out=hbuf, ctx=ctx)
                out = Kk.fused_quantized_linear(hid, *w2, False, ctx=ctx)
                assert np.array_equal(hid.numpy(), hid_ref), (b, m, rs)
                assert np.array_equal(out.numpy(), out_ref), (b, m, rs)
                # rewrite the buffer with different data of the same shape: the statistics must not survive
                y = Kk.mul(hid, np.array([0.5], np.float32), out=hbuf, ctx=ctx)
                assert np.array_equal(Kk.fused_quantized_linear(y, *w2, False, ctx=ctx).numpy(),
                                      orc.fused_quantized_linear(hid_ref * np.float32(0.5), w2[0].arr, w2[1].arr, [128.0], w2[3].arr, False)), (b, m, rs)


@pytest.mark.gpu
def test_prepared_weight_entry_points(ctx, orc):
    """prepare_weights / mat_mul_integer_prepared / fused_dq_gemm_prepared / mat_mul_integer_u8_weights
    (quantization.rs:173, 221, 454, 699) against the oracle's mat_mul_integer / fused_quantized_linear"""
    from lele_amd import kernels as Kk
    rng = np.random.default_rng(31)
    for b, m, k, n in ((1, 93, 512, 512), (2, 9, 37, 19), (3, 171, 512, 96)):
        wu8 = rng.integers(0, 256, (k, n), dtype=np.uint8)
        a = rng.integers(0, 256, (b, m, k)).astype(np.float32)
        ws = (np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)
        bias = (rng.standard_normal(n) * 0.02).astype(np.float32)
        pw = Kk.prepare_weights(wu8, k, n, ctx=ctx)
        want = orc.mat_mul_integer(a, wu8.astype(np.float32), [3.0], [131.0], ws, bias, relu=True)
        assert np.array_equal(Kk.mat_mul_integer_prepared(a, pw, 3.0, 131, ws, bias, True, ctx=ctx).numpy(), want)
        assert np.array_equal(Kk.mat_mul_integer_u8_weights(a, wu8, [k, n], 3.0, 131, ws, bias, True, ctx=ctx).numpy(), want)
        assert np.array_equal(Kk.mat_mul_integer_prepared(a, pw, None, None, None, None, False, ctx=ctx).numpy(),
                              orc.mat_mul_integer(a, wu8.astype(np.float32)))
        x = (rng.standard_normal((b, m, k)) * 2).astype(np.float32)
        assert np.array_equal(Kk.fused_dq_gemm_prepared(x, pw, 128, ws, bias, False, ctx=ctx).numpy(),
                              orc.fused_quantized_linear(x, wu8.astype(np.float32), ws, [128.0], bias, False))
        pw.close()


def _ln_params(rng, n=512):
    from lele_amd._lib import Weight
    return Weight((1 + 0.1 * rng.standard_normal(n)).astype(np.float32)), Weight((0.1 * rng.standard_normal(n)).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("b,m,n,nres", [
    (32, 171, 512, 2),    # one configs[3] shard: the LayerNorm runs in igemm_as_kernel's epilogue
    (32, 171, 512, 1), (32, 171, 512, 0),
    (3, 1000, 512, 2),    # slices that straddle row tiles, a ragged last tile (3000 rows = 93.75 tiles)
    (86, 32, 512, 1),     # the smallest batch whose row tiles fill the chip: every tile is one slice
    (8, 171, 512, 2),     # too few row tiles for whole rows per workgroup: the two calls
    (1, 504, 512, 2),     # one utterance: the two calls
    (32, 171, 256, 2),    # N != 512: the two calls
])
def test_projection_residuals_and_layer_norm_as_one_call(ctx, orc, b, m, n, nres):
    """lele_hip_fused_quantized_linear_residual_ln == fused_quantized_linear_residual -> layer_norm, bit for bit, on the route that
    normalises in the GEMM's epilogue and on every route that issues the two calls; the {min, max} pairs left beside the normalised
    result feed the next quantised linear exactly as the LayerNorm kernel's do; one case against the oracle's own sequence."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(b * 1000 + m + n + nres)
    x = (rng.standard_normal((b, m, 512)) * rng.uniform(0.3, 3.0, (b, 1, 1))).astype(np.float32)
    g0, b0 = _ln_params(rng)
    xn = K.layer_norm(ctx.buf().upload(x), g0, b0, -1, 1e-5, out=ctx.buf(), ctx=ctx)
    w, ws, bias = _weights(rng, 512, n)
    W = (Weight(w), Weight(ws), Weight(np.array([128.0], np.float32)), Weight(bias))
    r1 = rng.standard_normal((b, m, n)).astype(np.float32) if nres >= 1 else None
    r2 = rng.standard_normal((b, m, n)).astype(np.float32) if nres >= 2 else None
    g1, b1 = _ln_params(rng, n)
    lin = K.fused_quantized_linear_residual(xn, *W, False, r1, r2, ctx=ctx) if nres else K.fused_quantized_linear(xn, *W, False, ctx=ctx)
    want = lin.numpy().copy(), K.layer_norm(lin, g1, b1, -1, 1e-5, ctx=ctx).numpy().copy()
    for rep in range(3):      # repeated: a kernel that read LDS before its loads had landed would not repeat itself
        got = K.fused_quantized_linear_residual_ln(xn, *W, False, r1, r2, g1, b1, 1e-5, ctx=ctx)
        assert np.array_equal(got[0].numpy(), want[0]) and np.array_equal(got[1].numpy(), want[1]), (b, m, n, nres, rep)
    w2, ws2, bias2 = _weights(rng, n, 1024)
    W2 = (Weight(w2), Weight(ws2), Weight(np.array([128.0], np.float32)), Weight(bias2))
    seq_next = K.fused_quantized_linear(K.layer_norm(lin, g1, b1, -1, 1e-5, ctx=ctx), *W2, True, ctx=ctx).numpy()
    assert np.array_equal(K.fused_quantized_linear(got[1], *W2, True, ctx=ctx).numpy(), seq_next)
    if (b, m, nres) == (3, 1000, 2):
        o_lin = orc.fused_quantized_linear(xn.numpy(), w, ws, [128.0], bias) + r1 + r2
        assert np.array_equal(want[0], o_lin) and np.array_equal(want[1], orc.layer_norm(o_lin, g1.arr, b1.arr, -1, 1e-5))


@pytest.mark.gpu
@pytest.mark.parametrize("b,t,res2,fbias,pads,koff", [
    (32, 171, True, False, (5, 5), 1024),    # configs[3]: one launch (memory block + projection + Adds + LayerNorm)
    (32, 171, False, True, (5, 5), 1024),    # the first layer's form (no second residual), with an FSMN bias
    (40, 100, True, False, (5, 5), 0),       # utterances shorter than four tiles: every tile crosses an utterance boundary somewhere
    (90, 33, True, True, (5, 5), 512),       # utterances barely longer than a tile; a window that spans three utterances
    (5, 700, True, False, (5, 5), 1024),     # a ragged last tile
    (32, 171, True, False, (10, 0), 1024),   # causal padding: the three calls
    (1, 504, True, False, (5, 5), 1024),     # one utterance: the three calls
])
def test_sanm_out_block_is_the_three_calls_bit_for_bit(ctx, orc, b, t, res2, fbias, pads, koff):
    """lele_hip_sanm_out_block == depthwise_conv1d_tlc(add_input) -> fused_quantized_linear_residual -> layer_norm: the FSMN memory
    block computed inside the projection kernel from a window of v in LDS (taps outside the utterance skipped, the sequence's FMA
    order), rows normalised in the epilogue.  Three repetitions (the window arrives through direct-to-LDS loads that only a counted
    wait orders before their first use), one case against the oracle's sequence."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(b * 100 + t)
    av = (rng.standard_normal((b, t, 512)) * rng.uniform(0.3, 3.0, (b, 1, 1))).astype(np.float32)
    qkv = rng.standard_normal((b, t, 1536)).astype(np.float32)
    qkv_d, av_d = ctx.buf().upload(qkv), ctx.buf().upload(av)
    w, ws, bias = _weights(rng, 512, 512)
    W = (Weight(w), Weight(ws), Weight(np.array([128.0], np.float32)), Weight(bias))
    fw = Weight((rng.standard_normal((512, 1, 11)) / np.sqrt(11)).astype(np.float32))
    fb = Weight((rng.standard_normal(512) * 0.1).astype(np.float32)) if fbias else None
    r2 = rng.standard_normal((b, t, 512)).astype(np.float32) if res2 else None
    g1, b1 = _ln_params(rng)
    mem = K.depthwise_conv1d_tlc(qkv_d, fw, fb, pads[0], pads[1], False, koff, True, ctx=ctx)
    lin = K.fused_quantized_linear_residual(av_d, *W, False, mem, r2, ctx=ctx)
    want = lin.numpy().copy(), K.layer_norm(lin, g1, b1, -1, 1e-5, ctx=ctx).numpy().copy()
    for rep in range(3):
        got = K.sanm_out_block(av_d, *W, False, qkv_d, fw, fb, koff, pads[0], pads[1], r2, g1, b1, 1e-5, ctx=ctx)
        assert np.array_equal(got[0].numpy(), want[0]), (b, t, rep, float(np.abs(got[0].numpy() - want[0]).max()))
        assert np.array_equal(got[1].numpy(), want[1]), (b, t, rep)
    if (b, t) == (40, 100):
        from oracle import plan_ref
        o = plan_ref.PlanRef({"statements": [], "weights": {}, "outputs": [], "inputs": []}, {}).call(
            "sanm_out_block", [av, w, ws, np.array([128.0], np.float32), bias, False, qkv, fw.arr, None, koff, 5, 5, r2, g1.arr, b1.arr, 1e-5])
        # against the oracle's sequence the bar is the convolution's (1e-4: the oracle's conv1d restates lele's dot loops, the device's
        # depthwise kernels chain FMAs -- tests/test_conv_rnn.py), everything around it is exact
        from tests.parity import close_f32
        close_f32(want[0], o[0], 1e-4, "x1")
        close_f32(want[1], o[1], 1e-4, "layer_norm(x1)")


@pytest.mark.gpu
@pytest.mark.parametrize("b,m,nres", [(32, 171, 1), (32, 171, 2), (32, 171, 0), (3, 1000, 1), (86, 32, 1), (8, 171, 1), (1, 504, 1)])
def test_feed_forward_block_and_next_layer_norm_as_one_call(ctx, orc, b, m, nres):
    """lele_hip_fused_ffn_quantized_ln == fused_ffn_quantized -> layer_norm, bit for bit: on the route whose second product runs one row
    tile a workgroup with the weights streamed (igemm_ask_kernel) and normalises in the epilogue, and on the routes that issue the two
    calls; the statistics left beside the normalised result feed the next quantised linear; one case against the oracle's sequence."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(7 * b + m + nres)
    x = (rng.standard_normal((b, m, 512)) * rng.uniform(0.3, 3.0, (b, 1, 1))).astype(np.float32)
    g0, b0 = _ln_params(rng)
    xn = K.layer_norm(ctx.buf().upload(x), g0, b0, -1, 1e-5, out=ctx.buf(), ctx=ctx)
    w1, ws1, bias1 = _weights(rng, 512, 2048)
    w2, ws2, bias2 = _weights(rng, 2048, 512)
    W1 = (Weight(w1), Weight(ws1), Weight(np.array([128.0], np.float32)), Weight(bias1))
    W2 = (Weight(w2), Weight(ws2), Weight(np.array([127.0], np.float32)), Weight(bias2))
    r1 = rng.standard_normal((b, m, 512)).astype(np.float32) if nres >= 1 else None
    r2 = rng.standard_normal((b, m, 512)).astype(np.float32) if nres >= 2 else None
    g1, b1 = _ln_params(rng)
    y = K.fused_ffn_quantized(xn, *W1, *W2, False, r1, r2, ctx=ctx)
    want = y.numpy().copy(), K.layer_norm(y, g1, b1, -1, 1e-5, ctx=ctx).numpy().copy()
    for rep in range(3):
        got = K.fused_ffn_quantized_ln(xn, *W1, *W2, False, r1, r2, g1, b1, 1e-5, ctx=ctx)
        assert np.array_equal(got[0].numpy(), want[0]) and np.array_equal(got[1].numpy(), want[1]), (b, m, nres, rep)
    nxt = K.fused_quantized_linear(K.layer_norm(y, g1, b1, -1, 1e-5, ctx=ctx), *W1, True, ctx=ctx).numpy()
    assert np.array_equal(K.fused_quantized_linear(got[1], *W1, True, ctx=ctx).numpy(), nxt)
    if (b, m) == (3, 1000):
        h = orc.fused_quantized_linear(xn.numpy(), w1, ws1, [128.0], bias1, relu=True)
        o = orc.fused_quantized_linear(h, w2, ws2, [127.0], bias2) + r1
        assert np.array_equal(want[0], o) and np.array_equal(want[1], orc.layer_norm(o, g1.arr, b1.arr, -1, 1e-5))
