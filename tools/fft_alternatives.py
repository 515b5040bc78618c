#!/usr/bin/env python3
"""Evidence for DESIGN.md section 3.1: can ANY front-end that does not replay the reference's radix-2 arithmetic meet the parity
bar (|d log-mel| <= 1e-4 * max(|log-mel|, 1e-2), the bar of tests/test_frontend_gpu.py and __graft_entry__.smoke) against the
oracle of lele's SenseVoiceFrontend::compute (/root/reference/src/features/pipeline.rs:85-193, src/kernels/fft.rs:172-266)?

Every variant shares the oracle's f32 pre-processing bit for bit (x32768, sequential frame mean, pre-emphasis, Hann) and differs
only from the FFT on:
  exact     the spectrum in float64 (numpy.fft.rfft), power and mel sums in float64 with the oracle's f32 filter weights, then f32 ln.
            This is the limit of every "better" algorithm: whatever it differs from the oracle by IS the reference's own round-off.
  radix4    a radix-4 Stockham FFT in f32 with f32 twiddles (north_star's "Stockham radix-2/4"), f32 power, the oracle's sparse mel in f32.
  r4+dense  radix4 + mel as a dense [frames x 257] . [257 x 80] f32 product (north_star's "mel as a dense MFMA GEMM";
            f32 MFMA is an FMA chain in k order: emulated as such).
  r4+bf16x3 radix4 + the dense mel product on 3-term split-bf16 operands (hi*hi + hi*lo + lo*hi, f32 accumulation).
Inputs: the configs[1] synthetic utterance (two sines + 5 % noise, seed 0, 30 s) and real speech (tests/golden/zh.wav).
Writes a JSON histogram of the error relative to the bar per variant; run on the CPU (no GPU needed):

    python tools/fft_alternatives.py --out profiles/r03_fft_alternatives.json
"""
import argparse
import json
import os
import sys
import wave

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
f32 = np.float32


def preprocess(pcm):
    """frames [F, 512] f32, bit-identical with oracle/features.cpp:350-359 (elementwise f32 ops; the frame mean is a sequential sum)"""
    n = (len(pcm) - 400) // 160 + 1
    idx = np.arange(400)[None, :] + 160 * np.arange(n)[:, None]
    raw = pcm[idx] * f32(32768.0)
    s = np.zeros(n, f32)
    for j in range(400):  # sequential f32 sum, vectorised over frames
        s = s + raw[:, j]
    raw = raw - (s / f32(400.0))[:, None]
    out = raw.copy()
    out[:, 1:] = raw[:, 1:] - f32(0.97) * raw[:, :-1]
    from oracle import pyoracle as O
    out = out * O.hann_window(400)[None, :]
    fr = np.zeros((n, 512), f32)
    fr[:, :400] = out
    return fr


def stockham_radix4_f32(x):
    """complex FFT of length 512 = 4^4 * 2 over the last axis, f32 arithmetic throughout (radix-4 Stockham passes + one radix-2)"""
    n = x.shape[-1]
    re, im = x.astype(f32), np.zeros_like(x, dtype=f32)
    # decimation in frequency, Stockham autosort: y[q + s*(4p + r)] from x[q + s*(p + r*m)]
    s, length = 1, n
    while length >= 4 and length % 4 == 0 and length > 2:
        m = length // 4
        p = np.arange(m)
        ang = (-2.0 * np.pi * p / length)
        w1r, w1i = np.cos(ang).astype(f32), np.sin(ang).astype(f32)
        w2r, w2i = np.cos(2 * ang).astype(f32), np.sin(2 * ang).astype(f32)
        w3r, w3i = np.cos(3 * ang).astype(f32), np.sin(3 * ang).astype(f32)
        xr = re.reshape(re.shape[:-1] + (4, m, s))
        xi = im.reshape(im.shape[:-1] + (4, m, s))
        a0r, a1r, a2r, a3r = xr[..., 0, :, :], xr[..., 1, :, :], xr[..., 2, :, :], xr[..., 3, :, :]
        a0i, a1i, a2i, a3i = xi[..., 0, :, :], xi[..., 1, :, :], xi[..., 2, :, :], xi[..., 3, :, :]
        t0r, t0i = a0r + a2r, a0i + a2i
        t1r, t1i = a0r - a2r, a0i - a2i
        t2r, t2i = a1r + a3r, a1i + a3i
        t3r, t3i = a1i - a3i, a3r - a1r  # -i * (a1 - a3)
        y0r, y0i = t0r + t2r, t0i + t2i
        y1r, y1i = t1r + t3r, t1i + t3i
        y2r, y2i = t0r - t2r, t0i - t2i
        y3r, y3i = t1r - t3r, t1i - t3i

        def tw(yr, yi, wr, wi):
            wr, wi = wr[:, None], wi[:, None]
            return (yr * wr - yi * wi).astype(f32), (yr * wi + yi * wr).astype(f32)
        y1r, y1i = tw(y1r, y1i, w1r, w1i)
        y2r, y2i = tw(y2r, y2i, w2r, w2i)
        y3r, y3i = tw(y3r, y3i, w3r, w3i)
        re = np.stack([y0r, y1r, y2r, y3r], axis=-2).reshape(re.shape)   # [..., m, 4, s]
        im = np.stack([y0i, y1i, y2i, y3i], axis=-2).reshape(im.shape)
        s *= 4
        length = m
    if length == 2:
        xr = re.reshape(re.shape[:-1] + (2, 1, s))
        xi = im.reshape(im.shape[:-1] + (2, 1, s))
        re = np.stack([xr[..., 0, :, :] + xr[..., 1, :, :], xr[..., 0, :, :] - xr[..., 1, :, :]], axis=-2).reshape(re.shape)
        im = np.stack([xi[..., 0, :, :] + xi[..., 1, :, :], xi[..., 0, :, :] - xi[..., 1, :, :]], axis=-2).reshape(im.shape)
    return re, im


def bf16_split(a):
    """a = hi + lo with hi, lo representable in bfloat16 (round to nearest even on the top 16 bits)"""
    def rnd(v):
        u = v.astype(f32).view(np.uint32).astype(np.uint64)
        u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
        return u.astype(np.uint32).view(f32)
    hi = rnd(a)
    lo = rnd((a - hi).astype(f32))
    return hi, lo


def fma_chain_matmul(p, w):
    """[F, K] . [K, N] as the f32 MFMA computes it: acc = fma(p[:, k], w[k, :], acc) for k in order (float64 product rounded once)"""
    acc = np.zeros((p.shape[0], w.shape[1]), f32)
    for k in range(p.shape[1]):
        if not w[k].any():
            continue
        acc = (acc.astype(np.float64) + p[:, k:k + 1].astype(np.float64) * w[k:k + 1].astype(np.float64)).astype(f32)
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    from oracle import pyoracle as O
    O.build()
    bank = O.mel_filterbank(16000, 512, 80, 20.0).astype(f32)  # dense [80, 257] view of the reference's filters
    n = 480000
    rng = np.random.default_rng(0)
    t = np.arange(n) / 16000.0
    synth = (0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1000 * t) + 0.05 * rng.uniform(-1, 1, n)).astype(f32)
    w = wave.open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "zh.wav"))
    speech = (np.frombuffer(w.readframes(w.getnframes()), np.int16).astype(f32) / f32(32768.0))
    edges = [0, 0.01, 0.03, 0.1, 0.3, 1, 3, 10, 30, 100, 1e9]
    report = {"bar": "|d| <= 1e-4 * max(|log-mel|, 1e-2)  (tests/test_frontend_gpu.py)", "histogram_edges_in_units_of_the_bar": edges[:-1] + ["inf"],
              "inputs": {}}
    for name, pcm in (("configs[1] synthetic 30 s", synth), ("tests/golden/zh.wav (real speech, 5.6 s)", speech)):
        _, ref = O.frontend_compute(pcm, return_mel=True)
        fr = preprocess(pcm)
        assert fr.shape[0] == ref.shape[0]
        # sanity: the oracle's own rfft on these frames reproduces the oracle's log-mel bit for bit (the pre-processing is shared)
        variants = {}
        spec = np.fft.rfft(fr.astype(np.float64), axis=-1)
        pw = spec.real ** 2 + spec.imag ** 2
        variants["exact (float64 FFT, power, mel)"] = np.log(np.maximum(pw @ bank.T.astype(np.float64), 1e-5)).astype(f32)
        re, im = stockham_radix4_f32(fr)
        chk = np.abs((re[:, :257] + 1j * im[:, :257]) - spec).max() / np.abs(spec).max()
        assert chk < 1e-5, chk
        p32 = (re[:, :257] * re[:, :257] + im[:, :257] * im[:, :257]).astype(f32)
        sparse = np.stack([O.sparse_mel_apply(16000, 512, 80, 20.0, None, p32[i]) for i in range(p32.shape[0])]) if p32.shape[0] <= 4000 else None
        if sparse is not None:
            variants["radix-4 Stockham f32 + the oracle's sparse mel"] = np.log(np.maximum(sparse, f32(1e-5))).astype(f32)
        dense = fma_chain_matmul(p32, bank.T.copy())
        variants["radix-4 Stockham f32 + dense f32-MFMA mel"] = np.log(np.maximum(dense, f32(1e-5))).astype(f32)
        ph, pl = bf16_split(p32)
        wh, wl = bf16_split(bank.T.copy())
        acc = (ph.astype(np.float64) @ wh.astype(np.float64) + ph.astype(np.float64) @ wl.astype(np.float64) +
               pl.astype(np.float64) @ wh.astype(np.float64)).astype(f32)
        variants["radix-4 Stockham f32 + 3-term split-bf16 mel"] = np.log(np.maximum(acc, f32(1e-5))).astype(f32)
        rows = {}
        bar = 1e-4 * np.maximum(np.abs(ref), 1e-2)
        for vn, got in variants.items():
            e = np.abs(got.astype(np.float64) - ref.astype(np.float64)) / bar
            hist = np.histogram(e, bins=edges)[0]
            fr_fail = e.max(axis=1) > 1.0
            rows[vn] = {"values": int(e.size), "fraction_of_values_outside_the_bar": float((e > 1).mean()),
                        "fraction_of_frames_with_a_value_outside": float(fr_fail.mean()), "max_error_in_bars": float(e.max()),
                        "median_error_in_bars": float(np.median(e)), "histogram": [int(h) for h in hist]}
            print("%-44s %-52s outside %.4f%% of values, %.2f%% of frames, max %.1f bars" % (name[:44], vn, 100 * (e > 1).mean(), 100 * fr_fail.mean(), e.max()))
        # where the failures live: dynamic range of the frame's power spectrum
        e = np.abs(variants["exact (float64 FFT, power, mel)"].astype(np.float64) - ref) / bar
        dr = 10 * np.log10(pw.max(axis=1) / np.maximum(np.exp(ref.astype(np.float64)).min(axis=1), 1e-5))
        rows["note"] = {"spectral_dynamic_range_db_of_frames_failing_exact": [float(np.percentile(dr[e.max(axis=1) > 1], q)) for q in (5, 50, 95)] if (e.max(axis=1) > 1).any() else [],
                        "spectral_dynamic_range_db_of_all_frames": [float(np.percentile(dr, q)) for q in (5, 50, 95)]}
        report["inputs"][name] = rows
    if args.out:
        json.dump(report, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
