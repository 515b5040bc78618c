"""ctypes binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY (see oracle/oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package `lele_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_f32p = C.POINTER(C.c_float)
c_i64p = C.POINTER(C.c_int64)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_hz_to_mel_htk.restype = C.c_float
        _LIB.orc_hz_to_mel_htk.argtypes = [C.c_float]
        _LIB.orc_mel_to_hz_htk.restype = C.c_float
        _LIB.orc_mel_to_hz_htk.argtypes = [C.c_float]
        for name in ("orc_frontend_shape", "orc_frontend_compute", "orc_stft", "orc_stft_power"):
            getattr(_LIB, name).restype = C.c_int64
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def f(x):
    return C.c_float(float(x))


def i64(x):
    return C.c_int64(int(x))


# ---------------------------------------------------------------- features
def hann_window(n):
    out = np.empty(n, np.float32)
    lib().orc_hann_window(i64(n), _p(out))
    return out


def precompute_twiddles(n):
    twr = np.zeros(n, np.float32)
    twi = np.zeros(n, np.float32)
    br = np.zeros(n, np.int64)
    lib().orc_precompute_twiddles(i64(n), _p(twr), _p(twi), _p(br))
    return twr[: n - 1], twi[: n - 1], br


def rfft(x, mode=2):
    x = _f32(x)
    n = x.shape[0]
    re = np.empty(n // 2 + 1, np.float32)
    im = np.empty(n // 2 + 1, np.float32)
    lib().orc_rfft(_p(x), i64(n), _p(re), _p(im), C.c_int(mode))
    return re, im


def hz_to_mel_htk(hz):
    return lib().orc_hz_to_mel_htk(hz)


def mel_to_hz_htk(mel):
    return lib().orc_mel_to_hz_htk(mel)


def mel_filterbank(sr, n_fft, n_mels, f_min, f_max=None):
    out = np.empty((n_mels, n_fft // 2 + 1), np.float32)
    lib().orc_mel_filterbank(f(sr), i64(n_fft), i64(n_mels), f(f_min), f(-1.0 if f_max is None else f_max), _p(out))
    return out


def sparse_mel_apply(sr, n_fft, n_mels, f_min, f_max, power):
    power = _f32(power)
    out = np.empty(n_mels, np.float32)
    lib().orc_sparse_mel_apply(f(sr), i64(n_fft), i64(n_mels), f(f_min), f(-1.0 if f_max is None else f_max),
                               _p(power), _p(out))
    return out


def frontend_shape(pcm_len, sample_rate=16000, frame_length_ms=25.0, frame_shift_ms=10.0, lfr_n=6):
    nf = C.c_int64(0)
    t = lib().orc_frontend_shape(i64(pcm_len), i64(sample_rate), f(frame_length_ms), f(frame_shift_ms), i64(lfr_n),
                                 C.byref(nf))
    return int(t), int(nf.value)


def frontend_compute(pcm, sample_rate=16000, n_mels=80, frame_length_ms=25.0, frame_shift_ms=10.0, lfr_m=7, lfr_n=6,
                     return_mel=False):
    """SenseVoiceFrontend::compute.  Returns LFR features [T, n_mels*lfr_m] (shape (0,) if too short)."""
    pcm = _f32(pcm)
    t, nf = frontend_shape(pcm.shape[0], sample_rate, frame_length_ms, frame_shift_ms, lfr_n)
    if t == 0:
        e = np.zeros((0,), np.float32)
        return (e, e) if return_mel else e
    mel = np.empty((nf, n_mels), np.float32)
    out = np.empty((t, n_mels * lfr_m), np.float32)
    r = lib().orc_frontend_compute(_p(pcm), i64(pcm.shape[0]), i64(sample_rate), i64(n_mels), f(frame_length_ms),
                                   f(frame_shift_ms), i64(lfr_m), i64(lfr_n), _p(mel), _p(out))
    assert r == t
    return (out, mel) if return_mel else out


def lfr(x, m=7, n=6):
    x = _f32(x)
    t, d = x.shape
    out = np.empty(((t + n - 1) // n, d * m), np.float32)
    lib().orc_lfr(_p(x), i64(t), i64(d), i64(m), i64(n), _p(out))
    return out


def cmvn(x, eps=1e-5):
    x = _f32(x)
    t, d = x.shape[-2], x.shape[-1]
    out = np.empty_like(x)
    lib().orc_cmvn(_p(x), i64(t), i64(d), f(eps), _p(out))
    return out


def cmvn_apply_with_stats(x, mean, std, eps=1e-5):
    x, mean, std = _f32(x), _f32(mean), _f32(std)
    t, d = x.shape[-2], x.shape[-1]
    out = np.empty_like(x)
    lib().orc_cmvn_apply_with_stats(_p(x), i64(t), i64(d), f(eps), _p(mean), _p(std), _p(out))
    return out


def _stft_frames(length, hop, win_length):
    return 1 if length < win_length else (length - win_length) // hop + 1


def stft(signal, n_fft, hop, win_length, window=None):
    s = _f32(signal).reshape(-1)
    nfr = _stft_frames(s.shape[0], hop, win_length)
    out = np.empty((nfr, n_fft // 2 + 1, 2), np.float32)
    w = _f32(window) if window is not None else None
    lib().orc_stft(_p(s), i64(s.shape[0]), i64(n_fft), i64(hop), i64(win_length), _p(w) if w is not None else None,
                   _p(out))
    return out


def stft_power(signal, n_fft, hop, win_length, window=None):
    s = _f32(signal).reshape(-1)
    nfr = _stft_frames(s.shape[0], hop, win_length)
    out = np.empty((nfr, n_fft // 2 + 1), np.float32)
    w = _f32(window) if window is not None else None
    lib().orc_stft_power(_p(s), i64(s.shape[0]), i64(n_fft), i64(hop), i64(win_length),
                         _p(w) if w is not None else None, _p(out))
    return out
