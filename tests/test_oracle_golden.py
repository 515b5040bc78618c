"""Pins the CPU oracle against every known-answer vector the reference's own tests hold for the feature path
(tests/golden/reference_kats.json; each entry cites the reference #[test] it was transcribed from)."""
import json
import os

import numpy as np

K = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kats.json")))


def test_fft8_known_answer(orc):
    k = K["fft8_kat"]
    for mode in (0, 1, 2):
        re, im = orc.rfft(np.array(k["input"], np.float32), mode)
        assert np.abs(re - np.array(k["re"])).max() < k["tol"]
        assert np.abs(im - np.array(k["im"])).max() < k["tol"]


def test_fft_avx2_matches_scalar(orc):
    k = K["fft_avx2_vs_scalar"]
    state = k["lcg"]["seed"]

    def nxt():
        nonlocal state
        state = (state * k["lcg"]["mul"] + k["lcg"]["add"]) & 0xFFFFFFFF
        return np.float32(np.float32(state) / np.float32(0xFFFFFFFF)) * np.float32(2.0) - np.float32(1.0)

    for log_n in k["log_n"]:
        n = 1 << log_n
        x = np.array([nxt() for _ in range(n)], np.float32)
        re_s, im_s = orc.rfft(x, 1)
        re_a, im_a = orc.rfft(x, 2)
        assert np.abs(re_s - re_a).max() < k["tol"] and np.abs(im_s - im_a).max() < k["tol"]


def test_hann4(orc):
    k = K["hann4"]
    w = orc.hann_window(k["size"])
    assert w.shape == (4,) and np.abs(w - np.array(k["expected"])).max() < k["tol"]


def test_fft_impulse_and_dc(orc):
    k = K["fft_impulse"]
    re, im = orc.rfft(np.array(k["input"], np.float32), 2)
    # RealFft::process mirrors bins n/2+1.. from the conjugates (features/fft.rs:36-42)
    full_re = np.concatenate([re, re[1:-1][::-1]])
    full_im = np.concatenate([im, -im[1:-1][::-1]])
    assert np.abs(full_re - np.array(k["re"])).max() < k["tol"] and np.abs(full_im).max() < k["tol"]
    k = K["fft_dc"]
    re, im = orc.rfft(np.array(k["input"], np.float32), 2)
    full_re = np.concatenate([re, re[1:-1][::-1]])
    assert np.abs(full_re - np.array(k["re"])).max() < k["tol"]


def test_mel_htk(orc):
    k = K["mel_htk"]
    for hz, mel, tol in k["hz_to_mel"]:
        assert abs(orc.hz_to_mel_htk(hz) - mel) < tol
    assert abs(orc.mel_to_hz_htk(orc.hz_to_mel_htk(k["roundtrip_hz"])) - k["roundtrip_hz"]) < k["tol"]


def test_mel_filterbank_shape(orc):
    k = K["mel_filterbank_shape"]
    w = orc.mel_filterbank(k["sr"], k["n_fft"], k["n_mels"], k["f_min"])
    assert w.shape == (k["n_mels"], k["n_fft"] // 2 + 1) and w.sum() > 0


def test_sparse_bank_equals_dense(orc):
    # SparseMelBank is the trimmed dense bank (mel.rs:56-90): same sums up to the dropped exact zeros
    rng = np.random.default_rng(3)
    p = rng.uniform(0, 1e6, 257).astype(np.float32)
    dense = orc.mel_filterbank(16000.0, 512, 80, 20.0)
    ref = np.array([np.sum((dense[i].astype(np.float64) * p)) for i in range(80)])
    got = orc.sparse_mel_apply(16000.0, 512, 80, 20.0, None, p)
    assert np.allclose(got, ref, rtol=1e-5)


def test_cmvn_basic(orc):
    k = K["cmvn_basic"]
    out = orc.cmvn(np.array(k["data"], np.float32).reshape(k["shape"]))
    assert abs(out[:, 0].mean()) < 1e-5 and out[0, 0] < 0 and abs(out[1, 0]) < 1e-5 and out[2, 0] > 0


def test_stft_power_dc(orc):
    k = K["stft_power_dc"]
    r = orc.stft_power(np.ones(k["signal_ones"], np.float32), k["n_fft"], k["hop"], k["win"])
    assert (r[:, 0] > 1000.0).all() and (r[:, 2:] < 1.0).all()


def test_stft_vs_power(orc):
    k = K["stft_vs_power"]
    sig = np.sin(np.arange(k["len"], dtype=np.float32) * np.float32(k["sin_step"])).astype(np.float32)
    s = orc.stft(sig, k["n_fft"], k["hop"], k["win"])
    p = orc.stft_power(sig, k["n_fft"], k["hop"], k["win"])
    assert s.shape[0] == p.shape[0]
    assert np.abs(s[..., 0] ** 2 + s[..., 1] ** 2 - p).max() < k["tol"]


def test_stft_sinusoid_peak(orc):
    k = K["stft_sinusoid"]
    i = np.arange(k["len"], dtype=np.float32)
    sig = np.sin(np.float32(2.0 * np.pi) * np.float32(k["freq"]) * i / np.float32(k["sr"])).astype(np.float32)
    r = orc.stft_power(sig, k["n_fft"], k["hop"], k["n_fft"])
    nfreq = k["n_fft"] // 2 + 1
    fb = int(round(k["freq"] / k["sr"] * k["n_fft"]))
    row = r[r.shape[0] // 2]
    others = [row[i] for i in range(nfreq) if i != fb and (i < 5 or i > nfreq - 5)]
    assert row[fb] > 5.0 * max(others)


def test_fft_precomputed_vs_scalar_parseval_linearity(orc):
    k = K["fft_pre_vs_scalar"]
    x = np.sin(np.arange(k["n"], dtype=np.float32) * np.float32(k["sin_step"])).astype(np.float32)
    a = orc.rfft(x, 0)
    b = orc.rfft(x, 2)
    assert np.abs(a[0] - b[0]).max() < k["tol"] and np.abs(a[1] - b[1]).max() < k["tol"]
    k = K["fft_parseval"]
    n = k["n"]
    i = np.arange(n, dtype=np.float32)
    x = (np.sin(i * np.float32(0.1)) + np.cos(i * np.float32(0.05))).astype(np.float32)
    re, im = orc.rfft(x, 2)
    p = re * re + im * im
    p[1:-1] *= 2
    assert abs((x * x).sum() - p.sum() / n) / (x * x).sum() < k["tol"]
    k = K["fft_linearity"]
    n = k["n"]
    i = np.arange(n, dtype=np.float32)
    a, b = np.sin(i * np.float32(0.5)).astype(np.float32), np.cos(i * np.float32(0.3)).astype(np.float32)
    ab = (a + np.float32(k["scale"]) * b).astype(np.float32)
    fa, fb, fab = orc.rfft(a, 2), orc.rfft(b, 2), orc.rfft(ab, 2)
    for c in (0, 1):
        assert np.abs(fab[c] - (fa[c] + k["scale"] * fb[c])).max() < k["tol"]


def test_frontend_against_float64_model(orc):
    """Independent second opinion: numpy float64 implementation of pipeline.rs on strong-signal bins."""
    from conftest import synth_pcm
    x = synth_pcm(16000 * 3, 5)
    out, mel = orc.frontend_compute(x, return_mel=True)
    fl, hop = 400, 160
    nf = (len(x) - fl) // hop + 1
    assert mel.shape == (nf, 80) and out.shape == ((nf + 5) // 6, 560)
    idx = np.arange(fl)[None, :] + hop * np.arange(nf)[:, None]
    fr = x[idx].astype(np.float64) * 32768
    fr -= fr.mean(1, keepdims=True)
    fr[:, 1:] -= 0.97 * fr[:, :-1].copy()
    w = 0.5 * (1 - np.cos(2 * np.pi * np.arange(fl) / (fl - 1)))
    P = np.abs(np.fft.rfft(fr * w, 512)) ** 2
    M = np.log(np.maximum(P @ orc.mel_filterbank(16000.0, 512, 80, 20.0).astype(np.float64).T, 1e-5))
    # the f32 reference carries ~1e-3 of round-off in its weakest bins; 5e-3 absolute separates bugs from noise
    assert np.abs(M - mel).max() < 5e-3
    # LFR is a pure gather of mel rows
    t = out.shape[0]
    for i in (0, 1, t // 2, t - 1):
        for b in range(7):
            src = min(max(i * 6 + b - 3, 0), nf - 1)
            assert np.array_equal(out[i, b * 80:(b + 1) * 80], mel[src])


def test_frontend_short_input_is_empty(orc):
    assert orc.frontend_compute(np.zeros(399, np.float32)).shape == (0,)
    assert orc.frontend_compute(np.zeros(400, np.float32)).shape == (1, 560)
