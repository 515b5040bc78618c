"""lele::kernels on the device (host mirror of /root/reference/src/kernels/mod.rs:23-39 re-exports).

Each function keeps the reference's name and argument order; `out` (a _lib.Buf) plays the role of the
`out: &mut Vec<f32>` workspace buffer and may be omitted.
"""
import ctypes as C

import numpy as np

from . import _lib
from .features import _ctx, _op
from .tensor import TensorView, unwrap  # noqa: F401


def stft(input, n_fft, hop_length, win_length, window=None, out=None, ctx=None):  # math.rs:2304
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_stft(ctx._h, _lib.as_tensor(unwrap(input), keep), C.c_int64(n_fft),
                                        C.c_int64(hop_length), C.c_int64(win_length),
                                        _lib.as_tensor(unwrap(window), keep), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def stft_power_spectrum(input, n_fft, hop_length, win_length, window=None, out=None, ctx=None):  # math.rs:2372
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_stft_power_spectrum(
        ctx._h, _lib.as_tensor(unwrap(input), keep), C.c_int64(n_fft), C.c_int64(hop_length), C.c_int64(win_length),
        _lib.as_tensor(unwrap(window), keep), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))
