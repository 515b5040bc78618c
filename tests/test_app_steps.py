"""App-side pre/post-processing on the device: WAV payload -> f32 (audio.rs:52-73) and the
greedy arg-max of the logits (tokenizer.rs:50-61).  CPU: the numpy restatement on hand-checked cases; GPU: bit-exact."""
import numpy as np
import pytest

from oracle import npref


def test_numpy_restatement_hand_cases():
    # i16 little-endian: 0x0000 -> 0, 0x4000 -> 0.5, 0x8000 -> -1.0, 0x7fff -> 32767/32768
    b = bytes([0x00, 0x00, 0x00, 0x40, 0x00, 0x80, 0xff, 0x7f])
    assert np.array_equal(npref.wav_to_f32(b, 16, 1), np.array([0.0, 0.5, -1.0, 32767 / 32768], np.float32))
    assert np.array_equal(npref.wav_to_f32(b, 16, 2), np.array([0.25, (-1.0 + 32767 / 32768) / 2], np.float32))
    assert np.array_equal(npref.wav_to_f32(bytes([0, 128, 255]), 8, 1), np.array([-1.0, 0.0, 127 / 128], np.float32))
    # Iterator::max_by returns the last maximum
    assert npref.argmax_last(np.array([[1, 3, 3, 2], [5, 5, 5, 5], [0, -1, -2, -3]], np.float32)).tolist() == [2, 3, 0]


@pytest.mark.gpu
def test_device_wav_to_f32_bit_exact(ctx):
    import lele_amd
    from lele_amd import kernels as K
    rng = np.random.default_rng(0)
    payload = rng.integers(0, 256, 2 * 48001 + 1, dtype=np.uint8).tobytes()  # odd trailing byte is dropped (chunks_exact)
    for bits, ch in ((16, 1), (8, 1), (8, 2)):
        pl = payload if (bits, ch) != (8, 2) else payload[:-1]
        assert np.array_equal(K.wav_to_f32(pl, bits, ch, ctx=ctx).numpy(), npref.wav_to_f32(pl, bits, ch))
    even = payload[:4 * 24000]
    assert np.array_equal(K.wav_to_f32(even, 16, 2, ctx=ctx).numpy(), npref.wav_to_f32(even, 16, 2))
    with pytest.raises(lele_amd.LeleError, match="Unsupported bits per sample: 24"):
        K.wav_to_f32(payload, 24, 1, ctx=ctx)


@pytest.mark.gpu
def test_device_argmax_last_bit_exact(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 171, 25055)).astype(np.float32)
    x[0, 5, 100] = x[0, 5, 20000] = 50.0   # tie: the later index wins
    x[1, 7, :] = -1.25                     # all equal: last index
    got = K.argmax_last(x, ctx=ctx).numpy()
    assert got.dtype == np.int32 and got.shape == (3, 171)
    assert np.array_equal(got, npref.argmax_last(x))
    assert got[0, 5] == 20000 and got[1, 7] == 25054
    assert K.argmax_last(np.array([3.0, 1.0], np.float32), ctx=ctx).numpy().tolist() == 0


# ---- token filter, image pre-processing, segmentation post-processing, VAD segments (oracle/apps.cpp restates the reference)
def _vocab(v=200):
    toks = ["<blank>"] + ["t%d" % i for i in range(1, v)]
    for i in (1, 2, 24, 25, 199):
        toks[i] = "<|s%d|>" % i
    toks[30] = "▁hello"
    toks[31] = "▁world"
    return toks


def test_token_filter_and_detokenize_host_cases():
    from lele_amd import apps
    from oracle import pyoracle as O
    toks = _vocab()
    skip = apps.special_token_mask(toks)
    assert skip.sum() == 6 and skip[0] == 1 and skip[24] == 1 and skip[30] == 0
    logits = np.full((1, 6, 200), -5.0, np.float32)
    for t, i in enumerate([1, 30, 0, 31, 31, 199]):  # special, hello, blank, world, world (no CTC collapsing upstream), special
        logits[0, t, i] = 3.0
    ids, counts = O.decode_greedy_ids(logits, skip)
    assert counts.tolist() == [3] and ids[0].tolist() == [30, 31, 31, -1, -1, -1]
    assert apps.detokenize(ids[0], toks) == "hello world world"


def test_vad_segments_host_matches_oracle():
    from lele_amd import apps
    from oracle import pyoracle as O
    rng = np.random.default_rng(7)
    # hand case: 512-sample chunks at 16 kHz; speech in chunks 10..39 and 44..60 -> one merged segment (gap 4 chunks < 200 ms + pads)
    probs = np.zeros(100, np.float32)
    probs[10:40] = 0.9
    probs[44:61] = 0.8
    n = 100 * 512
    got = apps.vad_segments(probs, 512, n, n - 100)
    assert got == O.vad_segments(probs, 512, n, n - 100)
    assert got == [(10 * 512 - 1920, 68 * 512 + 1920)]  # pad 120 ms = 1920 samples; silence reaches 200 ms = 3200 samples at chunk 67
    for it in range(200):
        k = int(rng.integers(1, 400))
        probs = (rng.random(k) ** (1 + it % 3)).astype(np.float32)
        if it % 4 == 0:
            probs = np.repeat((rng.random(k // 8 + 1) > 0.5).astype(np.float32), 8)[:k] * 0.9
        chunk = int(rng.choice([256, 512, 1536]))
        padded = k * chunk
        audio = padded - int(rng.integers(0, chunk))
        kw = dict(sample_rate=int(rng.choice([8000, 16000])), threshold=float(rng.choice([0.3, 0.5])),
                  min_silence_ms=float(rng.choice([100.0, 200.0])), min_speech_ms=float(rng.choice([250.0, 400.0])),
                  speech_pad_ms=float(rng.choice([0.0, 30.0, 120.0])), merge_gap_ms=float(rng.choice([0.0, 200.0])))
        assert apps.vad_segments(probs, chunk, padded, audio, **kw) == O.vad_segments(probs, chunk, padded, audio, **kw), (it, kw)


def test_image_preprocess_oracle_hand_case():
    from oracle import pyoracle as O
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)  # H=2, W=3
    out = O.image_preprocess(img, 4)
    # x: floor((x+0.5)*3/4) = 0,1,1,2 ; y: floor((y+0.5)*2/4) = 0,0,1,1
    want = img[np.array([0, 0, 1, 1])][:, np.array([0, 1, 1, 2])].transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)
    assert np.array_equal(out[0], want)


def _seg_inputs(rng, n_keep=12, hm=160):
    logits = np.zeros((1, 300, 38), np.float32)
    logits[0, :, 4] = rng.random(300) * 0.2                       # low scores
    keep = rng.choice(300, n_keep, replace=False)
    for i in keep:
        x1, y1 = rng.random(2) * 400
        logits[0, i, :4] = [x1, y1, x1 + 40 + rng.random() * 200, y1 + 40 + rng.random() * 200]
        logits[0, i, 4] = 0.55 + rng.random() * 0.45
        logits[0, i, 5] = float(rng.integers(0, 90))              # some beyond 79: clamped
        logits[0, i, 6:] = rng.standard_normal(32)
    bad = [int(j) for j in range(300) if j not in set(keep.tolist())][0]
    logits[0, bad, :6] = [100, 100, 90, 200, 0.99, 3]               # inverted box: skipped although the score passes
    feat = rng.standard_normal((1, 32, hm, hm)).astype(np.float32) * 0.5
    return logits, feat


def test_yolo_postprocess_oracle_properties():
    from oracle import pyoracle as O
    logits, feat = _seg_inputs(np.random.default_rng(3))
    dets, mask = O.yolo_seg_postprocess(logits, feat, 500, 375, 0.5)
    assert dets.shape == (12, 38) and set(np.unique(mask)) <= {0, 255} and mask.any()
    assert (dets[:, 5] <= 79).all() and (dets[:, 2] <= 500).all() and (dets[:, 3] <= 375).all()
    ys, xs = np.nonzero(mask)   # every set pixel lies inside some kept box
    inside = np.zeros(len(ys), bool)
    for d in dets:
        inside |= (xs >= d[0]) & (xs <= d[2]) & (ys >= d[1]) & (ys <= d[3])
    assert inside.all()
    d0, m0 = O.yolo_seg_postprocess(logits, feat, 500, 375, 1.5)  # nothing passes: empty detections, zero mask
    assert d0.shape[0] == 0 and not m0.any()


@pytest.mark.gpu
def test_device_token_filter_bit_exact(ctx):
    from lele_amd import apps, kernels as K
    from oracle import pyoracle as O
    rng = np.random.default_rng(11)
    toks = _vocab(200)
    skip = apps.special_token_mask(toks)
    for b, t in ((1, 6), (3, 171), (2, 700), (4, 1)):
        logits = rng.standard_normal((b, t, 200)).astype(np.float32)
        logits[..., 0] += 1.5  # plenty of blanks
        ids = K.argmax_last(logits, ctx=ctx)
        got_ids, got_counts = K.token_filter(ids, skip, ctx=ctx)
        want_ids, want_counts = O.decode_greedy_ids(logits, skip)
        assert np.array_equal(got_counts.numpy(), want_counts) and np.array_equal(got_ids.numpy(), want_ids), (b, t)
    # ids beyond the vocabulary are dropped (tokenizer.rs:63)
    ids = np.array([[5, 250, 30, 0, 31]], np.int32)
    gi, gc = K.token_filter(ids, skip, ctx=ctx)
    assert gi.numpy().tolist() == [[5, 30, 31, -1, -1]] and gc.numpy().tolist() == [3]


@pytest.mark.gpu
def test_device_image_preprocess_bit_exact(ctx):
    from lele_amd import kernels as K
    from oracle import pyoracle as O
    rng = np.random.default_rng(5)
    for h, w, target in ((480, 640, 640), (375, 500, 640), (1080, 1920, 640), (17, 5, 32), (640, 640, 640)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = K.image_preprocess(img, target, ctx=ctx).numpy()
        assert got.shape == (1, 3, target, target) and np.array_equal(got, O.image_preprocess(img, target)), (h, w, target)


@pytest.mark.gpu
def test_device_yolo_postprocess_matches_oracle(ctx):
    from lele_amd import kernels as K
    from oracle import pyoracle as O
    for seed, (iw, ih), hm, thr in ((3, (500, 375), 160, 0.5), (4, (640, 640), 160, 0.6), (5, (1280, 720), 80, 0.5), (6, (33, 47), 16, 0.5)):
        logits, feat = _seg_inputs(np.random.default_rng(seed), hm=hm)
        dets, count, mask = K.yolo_seg_postprocess(logits, feat, iw, ih, thr, 80, ctx=ctx)
        want_dets, want_mask = O.yolo_seg_postprocess(logits, feat, iw, ih, thr)
        n = int(count.numpy()[0])
        assert n == want_dets.shape[0] and np.array_equal(dets.numpy()[:n], want_dets)      # bit-exact records, query order
        got_mask = mask.numpy()
        assert got_mask.shape == (ih, iw)
        # the mask sigmoid's expf is libm upstream and double-rounded here: identical except at 1-ulp ties at the 0.5 tests
        assert (got_mask != want_mask).mean() <= 1e-5, (seed, (got_mask != want_mask).sum())
    dets, count, mask = K.yolo_seg_postprocess(logits, feat, 64, 64, 1.5, 80, ctx=ctx)   # nothing passes
    assert int(count.numpy()[0]) == 0 and not mask.numpy().any()


@pytest.mark.gpu
def test_device_yolo_postprocess_over_a_batch_is_the_per_image_routine(ctx):
    """[N, 300, 38] + [N, 32, Hm, Wm] in one call == image.rs:127-265 image by image (the oracle), and the rows behind an image's kept
    detections are zeros: the fixed-width block the ranks of a sharded batch exchange ("C5")"""
    from lele_amd import kernels as K
    from oracle import pyoracle as O
    rng = np.random.default_rng(11)
    pairs = [_seg_inputs(rng, n_keep=4 + 5 * i, hm=40) for i in range(5)]
    logits = np.concatenate([p[0].reshape(1, 300, 38) for p in pairs])
    feat = np.concatenate([p[1].reshape(1, 32, 40, 40) for p in pairs])
    dets, count, mask = K.yolo_seg_postprocess(logits, feat, 96, 72, 0.5, 80, ctx=ctx)
    d, c, m = dets.numpy(), count.numpy(), mask.numpy()
    assert d.shape == (5, 300, 38) and c.shape == (5,) and m.shape == (5, 72, 96)
    for i, (lg, ft) in enumerate(pairs):
        want_dets, want_mask = O.yolo_seg_postprocess(lg, ft, 96, 72, 0.5)
        assert c[i] == want_dets.shape[0] and np.array_equal(d[i, :c[i]], want_dets) and not d[i, c[i]:].any(), i
        assert (m[i] != want_mask).mean() <= 1e-4, i
    assert len(set(c.tolist())) > 1          # the images really keep different numbers of rows
