"""GPU parity of the audio front-end (through the C ABI) against the CPU oracle.

Tolerance (BASELINE.json north_star): mel within 1e-4 relative f32.  The device path reproduces the reference's
roundings up to the power spectrum and mel sum, so the observed difference is a last-ulp logf difference; the
assertion below is |gpu - oracle| <= 1e-4*|oracle| + 1e-6 element-wise.  LFR/gather structure is bit-exact."""
import os

import numpy as np
import pytest

from conftest import synth_pcm

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-4, 1e-6


def close(a, b):
    return np.all(np.abs(a - b) <= RTOL * np.abs(b) + ATOL)


@pytest.fixture(scope="module")
def fe(ctx):
    from lele_amd.features import SenseVoiceFrontend
    return SenseVoiceFrontend(ctx=ctx)


@pytest.mark.parametrize("n", [400, 401, 559, 560, 561, 1359, 1360, 4000, 16000, 48123, 160000])
def test_logmel_and_lfr_match_oracle(fe, orc, n):
    x = synth_pcm(n, seed=n % 7)
    ref_out, ref_mel = orc.frontend_compute(x, return_mel=True)
    mel = fe.logmel(x).numpy()
    assert mel.shape == ref_mel.shape
    assert close(mel, ref_mel), float(np.abs(mel - ref_mel).max())
    out = fe.compute(x)
    assert out.shape == ref_out.shape
    assert close(out.numpy(), ref_out)


def test_short_input_is_empty(fe):
    assert fe.compute(np.zeros(399, np.float32)).shape == ()  # TensorView::empty(), pipeline.rs:70-72
    assert fe.compute(np.zeros(400, np.float32)).shape == (1, 560)


def test_constant_and_silent_input_hit_the_log_floor(fe, orc):
    for x in (np.zeros(3200, np.float32), np.full(3200, 0.01, np.float32)):  # the reference's own dummy input (main.rs:53)
        ref = orc.frontend_compute(x)
        assert close(fe.compute(x).numpy(), ref)


def test_high_dynamic_range_matches(fe, orc):
    # strong tone + DC + very weak noise: the case where any deviation from the reference's roundings shows
    n = 32000
    t = np.arange(n) / 16000.0
    x = (0.5 * np.sin(2 * np.pi * 440 * t) + 0.3 + 1e-4 * np.random.default_rng(1).uniform(-1, 1, n)).astype(np.float32)
    ref_out, ref_mel = orc.frontend_compute(x, return_mel=True)
    assert close(fe.logmel(x).numpy(), ref_mel)
    assert close(fe.compute(x).numpy(), ref_out)


def test_lfr_is_a_gather_of_logmel_rows(fe):
    x = synth_pcm(20000, 3)
    mel = fe.logmel(x).numpy()
    out = fe.compute(x).numpy()
    nf = mel.shape[0]
    for i in range(out.shape[0]):
        for b in range(7):
            src = min(max(i * 6 + b - 3, 0), nf - 1)
            assert np.array_equal(out[i, b * 80:(b + 1) * 80], mel[src])


def test_batch_equals_single_bit_for_bit(fe):
    xs = np.stack([synth_pcm(16000, s) for s in range(5)])
    got = fe.compute_batch(xs).numpy()
    for i in range(5):
        assert np.array_equal(got[i], fe.compute(xs[i]).numpy())


def test_full_size_30s_config2(fe, orc):
    x = synth_pcm(480000, 0)  # BASELINE config 2
    ref = orc.frontend_compute(x)
    got = fe.compute(x).numpy()
    assert got.shape == (500, 560) and close(got, ref)
    # size-independent property: time-shift by one hop shifts frames (frames 1.. of x == frames 0.. of x[160:])
    a = fe.logmel(x).numpy()
    b = fe.logmel(x[160:]).numpy()
    assert np.array_equal(a[1:], b)


lab_only = pytest.mark.skipif(os.environ.get("LELE_HIP_LAB") != "1",
                              reason="kernel-variant switches exist in the lab library only (LELE_HIP_LAB=1 python -m lele_amd.build)")


@lab_only
def test_shfl_variant_is_identical(ctx, fe):
    # default = DPP row_ror for the pre-emphasis neighbour; LELE_HIP_FE_DPP=0 selects the __shfl formulation
    from lele_amd.features import SenseVoiceFrontend
    os.environ["LELE_HIP_FE_DPP"] = "0"
    try:
        fe2 = SenseVoiceFrontend(ctx=ctx)
    finally:
        del os.environ["LELE_HIP_FE_DPP"]
    x = synth_pcm(16000, 11)
    assert np.array_equal(fe2.compute(x).numpy(), fe.compute(x).numpy())


@lab_only
@pytest.mark.parametrize("var", ["LELE_HIP_FE_FUSED", "LELE_HIP_FE_GENERIC_MEL"])
def test_kernel_variants_are_identical(ctx, fe, var):
    # LELE_HIP_FE_FUSED=0: separate fe_frame_sum_kernel + unfused main kernel (the first-half-of-round-1 form);
    # LELE_HIP_FE_GENERIC_MEL=1: table-driven mel loop instead of the unrolled default-bank rounds
    from lele_amd.features import SenseVoiceFrontend
    os.environ[var] = "0" if var.endswith("FUSED") else "1"
    try:
        fe2 = SenseVoiceFrontend(ctx=ctx)
    finally:
        del os.environ[var]
    xs = np.stack([synth_pcm(40000, s) for s in range(3)])
    assert np.array_equal(fe2.compute_batch(xs).numpy(), fe.compute_batch(xs).numpy())
    x = synth_pcm(12345, 2)  # unaligned length: scalar staging loads
    assert np.array_equal(fe2.compute(x).numpy(), fe.compute(x).numpy())
    assert np.array_equal(fe2.logmel(x).numpy(), fe.logmel(x).numpy())


def test_other_lfr_settings(ctx, orc):
    from lele_amd.features import FeatureConfig, SenseVoiceFrontend
    x = synth_pcm(9000, 4)
    for m, n, nm in ((7, 6, 80), (5, 3, 80), (1, 1, 40), (4, 7, 23)):
        f2 = SenseVoiceFrontend(FeatureConfig(n_mels=nm, lfr_m=m, lfr_n=n), ctx=ctx)
        ref = orc.frontend_compute(x, n_mels=nm, lfr_m=m, lfr_n=n)
        got = f2.compute(x).numpy()
        assert got.shape == ref.shape and close(got, ref)


@pytest.mark.parametrize("sr,fl,fs,nm", [(8000, 25.0, 10.0, 80), (22050, 25.0, 10.0, 80), (32000, 25.0, 10.0, 64),
                                         (16000, 20.0, 8.0, 40), (16000, 32.0, 16.0, 80), (40000, 25.0, 10.0, 128)])
def test_other_feature_configs_take_the_generic_path(ctx, orc, sr, fl, fs, nm):
    # any FeatureConfig the reference accepts (pipeline.rs:38-65): other framings run the composed bit-exact path
    from lele_amd.features import FeatureConfig, SenseVoiceFrontend
    x = synth_pcm(int(sr * 0.9), 5)
    f2 = SenseVoiceFrontend(FeatureConfig(sample_rate=sr, n_mels=nm, frame_length_ms=fl, frame_shift_ms=fs), ctx=ctx)
    ref, mel = orc.frontend_compute(x, sample_rate=sr, n_mels=nm, frame_length_ms=fl, frame_shift_ms=fs, return_mel=True)
    got = f2.compute(x).numpy()
    assert got.shape == ref.shape and close(got, ref)
    assert close(f2.logmel(x).numpy(), mel)
    xs = np.stack([synth_pcm(int(sr * 0.5), s) for s in range(3)])
    gb = f2.compute_batch(xs).numpy()
    for i in range(3):
        assert np.array_equal(gb[i], f2.compute(xs[i]).numpy())
    assert f2.compute(x[:10]).shape == ()  # shorter than one frame -> TensorView::empty()


@lab_only
@pytest.mark.parametrize("sr,fl,fs,nm", [(32000, 25.0, 10.0, 64), (16000, 20.0, 8.0, 40), (8000, 25.0, 10.0, 80), (40000, 25.0, 10.0, 128)])
def test_generic_fused_kernel_equals_the_four_kernels(ctx, sr, fl, fs, nm):
    """fe_generic_fused_kernel (frame sums, pre-emphasis + window, the radix-2 network and the sparse mel sums of FPB frames in one
    workgroup's LDS) against the four launches it replaces (LELE_HIP_FE_FUSED=0, lab library): the same operations in the same order,
    so every log-mel value and every LFR row bit for bit -- incl. a frame count that is not a multiple of the frames per workgroup"""
    from lele_amd.features import FeatureConfig, SenseVoiceFrontend
    cfg = FeatureConfig(sample_rate=sr, n_mels=nm, frame_length_ms=fl, frame_shift_ms=fs)
    one = SenseVoiceFrontend(cfg, ctx=ctx)
    os.environ["LELE_HIP_FE_FUSED"] = "0"
    try:
        four = SenseVoiceFrontend(cfg, ctx=ctx)
    finally:
        del os.environ["LELE_HIP_FE_FUSED"]
    xs = np.stack([synth_pcm(int(sr * 1.37), s) for s in range(3)])
    assert np.array_equal(one.logmel(xs).numpy(), four.logmel(xs).numpy())
    assert np.array_equal(one.compute_batch(xs).numpy(), four.compute_batch(xs).numpy())


def test_generic_path_batches_in_passes(ctx):
    """the generic path over a batch (one launch of the fused generic kernel for all utterances; passes only when the log-mel scratch of the
    batch exceeds 256 MiB): every utterance equals its own single-utterance call bit for bit, whichever workgroup / pass it fell into"""
    from lele_amd.features import FeatureConfig, SenseVoiceFrontend
    f2 = SenseVoiceFrontend(FeatureConfig(frame_length_ms=20.0, frame_shift_ms=8.0, n_mels=40), ctx=ctx)
    rng = np.random.default_rng(3)
    xs = (rng.standard_normal((24, 16000 * 30)) * 0.05).astype(np.float32)
    gb = f2.compute_batch(xs).numpy()
    lb = f2.logmel(xs).numpy()
    for i in (0, 11, 20, 21, 23):
        assert np.array_equal(gb[i], f2.compute(xs[i]).numpy()), i
        assert np.array_equal(lb[i], f2.logmel(xs[i]).numpy()), i


def test_unsupported_config_fails_loudly(ctx):
    import lele_amd
    from lele_amd.features import FeatureConfig, SenseVoiceFrontend
    # 48 kHz x 25 ms = 1200 samples > fft_len 1024: the reference indexes past frame_buf and panics (pipeline.rs:40,145)
    with pytest.raises(lele_amd.LeleError, match="do not fit fft_len"):
        SenseVoiceFrontend(FeatureConfig(sample_rate=48000), ctx=ctx)


# ---------------------------------------------------------------------------- small feature operators
def test_lfr_op_bit_exact(ctx, orc):
    from lele_amd.features import Lfr
    rng = np.random.default_rng(0)
    for t, d, m, n in ((1, 80, 7, 6), (5, 80, 7, 6), (6, 80, 7, 6), (7, 8, 7, 6), (100, 80, 7, 6), (33, 5, 3, 2)):
        x = rng.standard_normal((t, d)).astype(np.float32)
        assert np.array_equal(Lfr(m, n, ctx).compute(x).numpy(), orc.lfr(x, m, n))


def test_cmvn_bit_exact(ctx, orc):
    from lele_amd.features import Cmvn
    rng = np.random.default_rng(1)
    for t, d in ((1, 560), (3, 2), (93, 560), (500, 560)):
        x = (rng.standard_normal((t, d)) * 3 + 10).astype(np.float32)
        assert np.array_equal(Cmvn(ctx=ctx).compute(x).numpy(), orc.cmvn(x))
    x = rng.standard_normal((1, 17, 9)).astype(np.float32)
    assert np.array_equal(Cmvn(ctx=ctx).compute(x).numpy(), orc.cmvn(x))
    mean, std = rng.standard_normal(9).astype(np.float32), rng.uniform(0.5, 2, 9).astype(np.float32)
    assert np.array_equal(Cmvn(ctx=ctx).apply_with_stats(x, mean, std).numpy(),
                          orc.cmvn_apply_with_stats(x, mean, std))


def test_rfft_bit_exact(ctx, orc):
    from lele_amd.features import RealFft
    rng = np.random.default_rng(2)
    for n in (2, 4, 8, 16, 64, 512, 1024, 4096):
        x = rng.standard_normal((3, n)).astype(np.float32)
        re, im = RealFft(n, ctx).process(x)
        for r in range(3):
            ore, oim = orc.rfft(x[r], 2)
            assert np.array_equal(re.numpy()[r], ore) and np.array_equal(im.numpy()[r], oim)


def test_stft_ops_bit_exact(ctx, orc):
    from lele_amd import kernels as K
    sig = np.sin(np.arange(800, dtype=np.float32) * np.float32(0.01)).astype(np.float32)
    for shape in ((800,), (1, 800)):
        s = K.stft(sig.reshape(shape), 256, 128, 256, ctx=ctx)
        p = K.stft_power_spectrum(sig.reshape(shape), 256, 128, 256, ctx=ctx)
        assert s.shape == ((5, 129, 2) if len(shape) == 1 else (1, 5, 129, 2))
        assert np.array_equal(s.numpy().reshape(5, 129, 2), orc.stft(sig, 256, 128, 256))
        assert np.array_equal(p.numpy().reshape(5, 129), orc.stft_power(sig, 256, 128, 256))
    w = np.hanning(200).astype(np.float32)
    assert np.array_equal(K.stft(sig, 256, 64, 200, w, ctx=ctx).numpy(), orc.stft(sig, 256, 64, 200, w))
    short = sig[:100]
    assert np.array_equal(K.stft_power_spectrum(short, 256, 64, 256, ctx=ctx).numpy(),
                          orc.stft_power(short, 256, 64, 256))
