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
