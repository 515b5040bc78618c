"""The oracle's AVX2 inner loops (oracle/fast.cpp) against its own plain loops: identical bits.

oracle/fast.cpp restates the integer GEMM the way lele's x86 path evaluates it (vpmaddwd over a transposed, XOR-0x80
weight matrix with row/column-sum zero-point corrections, /root/reference/src/kernels/avx/quantization.rs:1203-1600) and
vectorises the k-ordered f32 FMA chain.  Both are exact re-orderings of independent lanes, so they must reproduce the
plain triple loops bit for bit -- which is what lets the fast form serve as the timed CPU baseline (bench.py) and as the
checker at full layer sizes."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as O


@pytest.fixture
def plain_flag():
    flag = C.c_int.in_dll(O.lib(), "orc_plain_loops")
    yield flag
    flag.value = 0


@pytest.mark.parametrize("m,k,n", [(1, 1, 1), (5, 7, 3), (33, 100, 37), (64, 512, 128), (3, 2048, 9), (2, 31, 4), (7, 33, 5)])
def test_int_gemm_fast_equals_plain(plain_flag, m, k, n):
    rng = np.random.default_rng(m * 1000 + k + n)
    x = rng.standard_normal((2, m, k)).astype(np.float32) * 3
    w = np.clip(np.round(128 + 80 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)
    s = (np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)
    b = (rng.standard_normal(n) * 0.02).astype(np.float32)
    for zb in (128.0, 0.0, 255.0, 3.0):
        z = np.array([zb], np.float32)
        plain_flag.value = 0
        fast = O.fused_quantized_linear(x, w, s, z, b, True)
        plain_flag.value = 1
        plain = O.fused_quantized_linear(x, w, s, z, b, True)
        assert np.array_equal(fast, plain)
    # mat_mul_integer: extreme operands (all 255 against zero points 0) exercise the widest accumulators
    a = np.full((1, m, k), 255.0, np.float32)
    bb = np.full((k, n), 255.0, np.float32)
    plain_flag.value = 0
    fast = O.mat_mul_integer(a, bb, np.array([0.0], np.float32), np.array([0.0], np.float32))
    plain_flag.value = 1
    plain = O.mat_mul_integer(a, bb, np.array([0.0], np.float32), np.array([0.0], np.float32))
    assert np.array_equal(fast, plain) and fast[0, 0, 0] == np.float32(255 * 255 * k)


@pytest.mark.parametrize("m,k,n", [(1, 1, 1), (5, 7, 3), (33, 100, 37), (19, 128, 48), (4, 3, 16), (6, 9, 17)])
def test_sgemm_kordered_fast_equals_plain(plain_flag, m, k, n):
    rng = np.random.default_rng(m + 10 * k + 100 * n)
    a = rng.standard_normal((3, m, k)).astype(np.float32)
    b = rng.standard_normal((3, k, n)).astype(np.float32)
    plain_flag.value = 0
    fast = O.matmul(a, b, acc32=True)
    plain_flag.value = 1
    plain = O.matmul(a, b, acc32=True)
    assert np.array_equal(fast, plain)
    ref = O.matmul(a, b)  # f64-accumulated
    assert np.allclose(fast, ref, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("geom", [(2, 16, 20, 24, 24, 3, 1, 1, [1, 1, 1, 1]), (1, 8, 17, 19, 12, 3, 2, 1, [1, 1, 1, 1]), (2, 12, 9, 9, 8, 1, 1, 1, [0, 0, 0, 0]),
                                  (1, 8, 16, 16, 8, 3, 1, 8, [1, 1, 1, 1]), (1, 12, 15, 15, 6, 3, 1, 2, [1, 2, 0, 1]), (1, 4, 12, 12, 6, 3, 1, 1, [0, 0, 2, 2]),
                                  (1, 6, 11, 13, 5, 5, 2, 1, [2, 2, 2, 2]), (1, 3, 33, 31, 16, 3, 2, 1, [1, 1, 1, 1])])
def test_conv2d_im2col_route_equals_the_float64_loop(geom):
    """oracle/conv_fast.cpp (lele's im2col + GEMM + bias / activation route, conv2d.rs:597-760, 892-1046) against orc_conv2d's
    float64-accumulated direct loop: f32 round-off apart, including a bias of exactly 0.0 (whose pass the reference skips) and
    every activation"""
    from parity import close_f32
    n, c, h, w_, oc, k, s, g, pads = geom
    rng = np.random.default_rng(sum(geom[:8]))
    x = rng.standard_normal((n, c, h, w_)).astype(np.float32)
    w = rng.standard_normal((oc, c // g, k, k)).astype(np.float32)
    b = rng.standard_normal(oc).astype(np.float32)
    b[0] = 0.0
    for act in (None, "relu", "silu"):
        for bias in (b, None):
            close_f32(O.conv2d_im2col(x, w, bias, [1, 1], g, pads, [s, s], act), O.conv2d(x, w, bias, [1, 1], g, pads, [s, s], act), 2e-6, str(geom))
    close_f32(O.conv2d_im2col(x, w, b, [2, 1], g, pads, [s, s], None), O.conv2d(x, w, b, [2, 1], g, pads, [s, s], None), 2e-6, "dilated " + str(geom))
