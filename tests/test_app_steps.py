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
