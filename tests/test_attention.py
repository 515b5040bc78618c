"""lele_hip_attention_view: softmax(Q K^T * scale) V in one launch (emitted by lele_amd.compiler in place of
matmul_view -> softmax_scaled -> matmul_view).  Two kinds of check:

(a) the FINAL output against the oracle's three-operator composition (matmul -> softmax -> matmul, each inside 1e-4 of the reference
    by its own test) at 2e-4, and against the three-call sequence the kernel replaces at 1e-4
    (test_attention_view_against_oracle_and_sequence, test_one_pass_attention_of_a_batch_against_the_oracle);
(b) OPERATOR BY OPERATOR at the north_star bar of 1e-4, through the fused kernel itself -- it has no taps, so the operands are chosen
    to make it hand an intermediate back (test_fused_attention_operator_by_operator):
      * V = one-hot rows (a 128-key chunk at a time): O IS the probability matrix P, exactly -- P against
        orc.softmax(orc.matmul(q, kT) * scale): the score product and the softmax;
      * K^T = 2^k I: the scores ARE 2^k Q, exactly -- P read back the same way against orc.softmax of that: the softmax stage alone;
      * Q = 0: P is uniform -- O = mean over the keys of V against the f64 mean: the P V product alone."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

H, DH = 4, 128
QC = [["slice", 2, 0, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]
KC = [["slice", 2, 512, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 3, 1]]]
VC = [["slice", 2, 1024, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]


class _env:
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


def close(a, b, what, rtol=1e-4):
    b = np.asarray(b, np.float32)
    floor = rtol * float(np.sqrt(np.mean(np.square(b, dtype=np.float64)))) + 1e-7
    bad = np.abs(a - b) > rtol * np.abs(b) + floor
    assert not bad.any(), "%s: %d of %d elements outside %g (max abs diff %.3g)" % (what, int(bad.sum()), b.size, rtol, float(np.abs(a - b).max()))


def close_p(a, b, what, rtol=1e-4, above=1e-4):
    """probabilities: `close` holds small ones only against rtol x rms; here every p >= `above` is ALSO held relatively on its own,
    |d| <= rtol p + 1e-7 (a floor of rtol x rms(P) says nothing about the relative error of a probability of 1e-3)"""
    close(a, b, what, rtol)
    b = np.asarray(b, np.float32)
    big = b >= above
    bad = big & (np.abs(a - b) > rtol * np.abs(b) + 1e-7)
    assert not bad.any(), "%s: %d of %d probabilities >= %g outside %g relative (worst %.3g)" % (
        what, int(bad.sum()), int(big.sum()), above, rtol, float((np.abs(a - b)[big] / b[big]).max()))


# the library picks the kernel by grid size: 16 query rows per workgroup for one utterance ((1, 504), (2, 33) ...), 32 for a few
# ((8, 171)), the one-pass batch kernel from half a chip's worth of 128-row blocks ((32, 171)); with LELE_HIP_ATTENTION_EXACT=1 the
# f32 replicas: 16 / 32 rows, and 64 rows per workgroup for large grids ((64, 171))
@pytest.mark.parametrize("exact", [0, 1])
@pytest.mark.parametrize("b,t", [(32, 171), (1, 504), (2, 33), (3, 64), (1, 512), (2, 100), (5, 1), (1, 8), (2, 65), (8, 171), (64, 171),
                                 (1, 300), (2, 257), (1, 256), (1, 449)])   # the last four: more single-utterance lengths (key counts off every tile boundary)
def test_attention_view_against_oracle_and_sequence(ctx, orc, b, t, exact):
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(b * 1000 + t)
    qkv = (rng.standard_normal((b, t, 3 * H * DH)) * 1.5).astype(np.float32)
    scale = Weight(np.array([DH ** -0.5], np.float32))
    qd = ctx.buf().upload(qkv)
    env = dict(LELE_HIP_ATTENTION_MIN_BLOCKS=1, LELE_HIP_ATTENTION_EXACT=exact)
    with _env(**env):   # the one-launch kernel whatever the grid size
        got = K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, H * DH], ctx=ctx)
    assert got.shape == (b, t, H * DH)
    got = got.numpy()
    # the oracle, operator by operator
    q = np.ascontiguousarray(qkv[..., :512].reshape(b, t, H, DH).transpose(0, 2, 1, 3))
    kT = np.ascontiguousarray(qkv[..., 512:1024].reshape(b, t, H, DH).transpose(0, 2, 3, 1))
    v = np.ascontiguousarray(qkv[..., 1024:].reshape(b, t, H, DH).transpose(0, 2, 1, 3))
    s = orc.matmul(q, kT) * np.float32(DH ** -0.5)
    p = orc.softmax(s, -1)
    o = np.ascontiguousarray(orc.matmul(p, v).transpose(0, 2, 1, 3)).reshape(b, t, H * DH)
    close(got, o, "fused attention vs oracle composition", rtol=2e-4)   # three operators deep: each inside 1e-4
    # the sequence it replaces
    with _env(LELE_HIP_ATTENTION_FUSED=0):
        seq = K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, H * DH], ctx=ctx).numpy()
    close(got, seq, "fused attention vs matmul_view -> softmax_scaled -> matmul_view")
    close(seq, o, "the sequence vs oracle composition", rtol=2e-4)


@pytest.mark.parametrize("b,t", [(32, 171), (40, 33), (33, 100), (130, 1), (16, 512), (35, 65), (40, 8), (32, 96), (32, 97)])
def test_one_pass_attention_of_a_batch_against_the_oracle(ctx, orc, b, t):
    """grids of >= 128 workgroups take attention_flash_kernel (loader wave + LDS ring, S^T / O^T in registers, online softmax,
    split-bf16 products): against the oracle's operator composition at the same 2e-4 as the other forms, and against the
    LELE_HIP_ATTENTION_EXACT=1 replica (f32 MFMA, the reference's row softmax) -- edge cases: one key, keys not a multiple of 32,
    exactly / one more than a workgroup's 96 query rows, 512 keys (the maximum), row blocks with nothing to do"""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(b * 77 + t)
    qkv = (rng.standard_normal((b, t, 3 * H * DH)) * 1.5).astype(np.float32)
    scale = Weight(np.array([DH ** -0.5], np.float32))
    qd = ctx.buf().upload(qkv)
    got = K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, H * DH], ctx=ctx).numpy()
    q = np.ascontiguousarray(qkv[..., :512].reshape(b, t, H, DH).transpose(0, 2, 1, 3))
    kT = np.ascontiguousarray(qkv[..., 512:1024].reshape(b, t, H, DH).transpose(0, 2, 3, 1))
    v = np.ascontiguousarray(qkv[..., 1024:].reshape(b, t, H, DH).transpose(0, 2, 1, 3))
    p = orc.softmax(orc.matmul(q, kT) * np.float32(DH ** -0.5), -1)
    o = np.ascontiguousarray(orc.matmul(p, v).transpose(0, 2, 1, 3)).reshape(b, t, H * DH)
    close(got, o, "one-pass attention vs oracle composition", rtol=2e-4)
    with _env(LELE_HIP_ATTENTION_EXACT=1):
        ex = K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, H * DH], ctx=ctx).numpy()
    close(got, ex, "one-pass attention vs the f32 replica kernel")
    close(ex, o, "replica vs oracle composition", rtol=2e-4)
    # no scale operand: plain softmax(Q K^T) V
    got1 = K.attention_view(qd, QC, qd, KC, qd, VC, None, [0, 2, 1, 3], [0, 0, H * DH], ctx=ctx).numpy()
    p1 = orc.softmax(orc.matmul(q, kT), -1)
    o1 = np.ascontiguousarray(orc.matmul(p1, v).transpose(0, 2, 1, 3)).reshape(b, t, H * DH)
    bad = np.abs(got1 - o1) > 2e-4 * np.abs(o1) + 2e-4 * float(np.sqrt(np.mean(np.square(o1, dtype=np.float64)))) + 1e-7
    assert bad.mean() <= 1e-5 and np.abs(got1 - o1).max() <= 1e-2   # unscaled scores of |s| ~ 30: near-ties amplify f32 round-off (see test_fullsize_graph)


def test_attention_statistics_feed_the_output_projection(ctx, orc):
    """the kernel leaves {min, max} per (utterance, head, row block) next to its result; the quantised output projection that
    reads it derives its per-utterance range from them -- same bits as the projection of a host copy (own range pass)"""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(17)
    w = (Weight(np.clip(np.round(128 + 32 * rng.standard_normal((512, 512))), 0, 255).astype(np.float32)),
         Weight((np.abs(rng.standard_normal(512)) * 0.01 + 0.002).astype(np.float32)), Weight(np.array([128.0], np.float32)),
         Weight((rng.standard_normal(512) * 0.02).astype(np.float32)))
    scale = Weight(np.array([DH ** -0.5], np.float32))
    for b, t in ((32, 171), (1, 504), (3, 40)):
        qkv = (rng.standard_normal((b, t, 1536)) * rng.uniform(0.5, 2.0, (b, 1, 1))).astype(np.float32)
        qd = ctx.buf().upload(qkv)
        with _env(LELE_HIP_ATTENTION_MIN_BLOCKS=1):
            av = K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, 512], ctx=ctx)
        got = K.fused_quantized_linear(av, *w, False, ctx=ctx).numpy()
        want = orc.fused_quantized_linear(av.numpy(), w[0].arr, w[1].arr, [128.0], w[3].arr, False)
        assert np.array_equal(got, want), (b, t)


def test_attention_view_other_geometries_run_the_sequence(ctx, orc):
    """head dimension 64 / more than 512 keys / a product with no scale: the call runs the three-call sequence"""
    from lele_amd import kernels as K
    rng = np.random.default_rng(3)
    q = rng.standard_normal((2, 3, 50, 64)).astype(np.float32)
    kT = rng.standard_normal((2, 3, 64, 70)).astype(np.float32)
    v = rng.standard_normal((2, 3, 70, 64)).astype(np.float32)
    got = K.attention_view(q, [], kT, [], v, [], None, ctx=ctx).numpy()
    o = orc.matmul(orc.softmax(orc.matmul(q, kT), -1), v)
    close(got, o, "sequence fallback", rtol=2e-4)


def _views(b, t, q, k, v):
    """separate [b, t, 512] buffers for Q, K, V as the views the compiler hands the kernel"""
    qc = [["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]
    kc = [["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 3, 1]]]
    return (q, qc, k, kc, v, qc)


@pytest.mark.parametrize("exact", [0, 1])
@pytest.mark.parametrize("b,t", [(32, 171), (1, 504), (64, 171), (16, 512), (8, 171)])
def test_fused_attention_operator_by_operator(ctx, orc, b, t, exact):
    """see the module docstring, (b).  (32, 171) / (64, 171) / (16, 512): attention_flash_kernel (with EXACT: the 64- / 32-row f32
    replicas); (1, 504): attention16_kernel; (8, 171): attention_kernel.  Every comparison at 1e-4."""
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(b * 31 + t)
    scale_v = np.float32(DH ** -0.5)
    scale = Weight(np.array([scale_v], np.float32))
    q = (rng.standard_normal((b, t, H * DH)) * 1.5).astype(np.float32)
    k = (rng.standard_normal((b, t, H * DH)) * 1.5).astype(np.float32)
    v = (rng.standard_normal((b, t, H * DH)) * 1.5).astype(np.float32)
    qh = np.ascontiguousarray(q.reshape(b, t, H, DH).transpose(0, 2, 1, 3))
    kT = np.ascontiguousarray(k.reshape(b, t, H, DH).transpose(0, 2, 3, 1))
    vh = np.ascontiguousarray(v.reshape(b, t, H, DH).transpose(0, 2, 1, 3))
    qd, kd = ctx.buf().upload(q), ctx.buf().upload(k)

    def run(qb, kb, vb):
        with _env(LELE_HIP_ATTENTION_MIN_BLOCKS=1, LELE_HIP_ATTENTION_EXACT=exact):
            return K.attention_view(*_views(b, t, qb, kb, vb), scale, None, None, ctx=ctx).numpy()      # [b, H, t, DH]

    def read_p(qb, kb):
        """the kernel's probability matrix [b, H, t, t], read back 128 keys at a time through one-hot value rows"""
        p = np.zeros((b, H, t, t), np.float32)
        for c0 in range(0, t, DH):
            n = min(DH, t - c0)
            one_hot = np.zeros((b, t, H, DH), np.float32)
            one_hot[:, c0 + np.arange(n), :, np.arange(n)] = 1.0        # value row of key c0 + d = e_d, in every head
            o = run(qb, kb, ctx.buf().upload(one_hot.reshape(b, t, H * DH)))
            p[..., c0:c0 + n] = o[..., :n]
            assert not o[..., n:].any() or n == DH
        return p
    # 1. score product + softmax
    p_dev = read_p(qd, kd)
    p_ref = orc.softmax(orc.matmul(qh, kT) * scale_v, -1)
    close_p(p_dev, p_ref, "P = softmax(Q K^T s) read back through one-hot V")
    assert np.abs(p_dev.sum(-1) - 1).max() < 1e-5
    # 2. the softmax stage alone: K^T = 4 I (its first 128 keys; the others zero) -> scores = 4 Q exactly (0 beyond key 128)
    k_id = np.zeros((b, t, H, DH), np.float32)
    n = min(DH, t)
    k_id[:, np.arange(n), :, np.arange(n)] = 4.0
    s_exact = np.zeros((b, H, t, t), np.float32)
    s_exact[..., :n] = 4.0 * qh[..., :n]
    p_dev2 = read_p(qd, ctx.buf().upload(k_id.reshape(b, t, H * DH)))
    close_p(p_dev2, orc.softmax(s_exact * scale_v, -1), "softmax of exactly known scores")
    # 3. the P V product alone: Q = 0 -> P = 1 / t everywhere -> O = mean over keys of V
    o3 = run(ctx.buf().upload(np.zeros_like(q)), kd, ctx.buf().upload(v))
    want = np.broadcast_to(vh.astype(np.float64).mean(axis=2, keepdims=True), vh.shape)
    close(o3, want.astype(np.float32), "uniform P times V against the f64 mean")
