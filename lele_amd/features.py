"""lele::features on the device (host mirror of /root/reference/src/features/*.rs)."""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .tensor import TensorView, unwrap


@dataclass
class FeatureConfig:  # pipeline.rs:8-27
    sample_rate: int = 16000
    n_mels: int = 80
    frame_length_ms: float = 25.0
    frame_shift_ms: float = 10.0
    lfr_m: int = 7
    lfr_n: int = 6


def _ctx(ctx):
    from . import default_ctx
    return ctx if ctx is not None else default_ctx()


class SenseVoiceFrontend:
    """SenseVoiceFrontend (pipeline.rs:28-193): compute(pcm) -> TensorView [T, n_mels*lfr_m]."""

    def __init__(self, config=None, ctx=None):
        self.config = config or FeatureConfig()
        self.ctx = _ctx(ctx)
        c = self.config
        cfg = _lib.LeleFeatureConfig(c.sample_rate, c.n_mels, c.frame_length_ms, c.frame_shift_ms, c.lfr_m, c.lfr_n)
        self._h = C.c_void_p()
        _lib.check(_lib.lib().lele_hip_frontend_create(self.ctx._h, C.byref(cfg), C.byref(self._h)))
        self._out = self.ctx.buf()

    def __del__(self):
        try:
            if self._h:
                _lib.lib().lele_hip_frontend_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def out_rows(self, pcm_len):
        rows, cols, nf = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(_lib.lib().lele_hip_frontend_out_rows(self._h, C.c_int64(pcm_len), C.byref(rows), C.byref(cols),
                                                         C.byref(nf)))
        return rows.value, cols.value, nf.value

    def set_profiling(self, on):
        _lib.check(_lib.lib().lele_hip_frontend_set_profiling(self._h, C.c_int(1 if on else 0)))

    def profile_read(self):
        """-> (avg ms of fe_frame_sum_kernel, avg ms of fe_main_kernel, runs)"""
        a, b, n = C.c_float(), C.c_float(), C.c_int64()
        _lib.check(_lib.lib().lele_hip_frontend_profile_read(self._h, C.byref(a), C.byref(b), C.byref(n)))
        return a.value, b.value, n.value

    def _call(self, fn, pcm, out):
        keep = []
        t = _lib.as_tensor(unwrap(pcm), keep)
        sh = _lib.OutShape()
        out = out or self._out
        _lib.check(fn(self._h, t, out._h, sh.shape, C.byref(sh.rank)))
        if sh.rank.value == 0:
            return TensorView.empty()  # pcm shorter than one frame (pipeline.rs:70-72)
        return TensorView(_lib.DevTensor(out, sh.get(), np.float32))

    def compute(self, pcm, out=None):
        return self._call(_lib.lib().lele_hip_frontend_compute, pcm, out)

    def compute_batch(self, pcm, out=None):
        """pcm [batch, pcm_len] -> [batch, T, n_mels*lfr_m] (one launch pair for the whole batch)"""
        return self._call(_lib.lib().lele_hip_frontend_compute_batch, pcm, out)

    def logmel(self, pcm, out=None):
        """intermediate log-mel [num_frames, n_mels] (before LFR); test hook"""
        return self._call(_lib.lib().lele_hip_frontend_logmel, pcm, out)


def _op(ctx, fn, tensors, extra, out=None, dtype=np.float32, prefix=()):
    ctx = _ctx(ctx)
    keep = []
    args = [ctx._h] + list(prefix)
    for t in tensors:
        args.append(_lib.as_tensor(unwrap(t), keep))
    args.extend(extra)
    out = out or ctx.buf()
    sh = _lib.OutShape()
    args.extend([out._h, sh.shape, C.byref(sh.rank)])
    _lib.check(fn(*args))
    return TensorView(_lib.DevTensor(out, sh.get(), dtype))


class Lfr:  # lfr.rs
    def __init__(self, m=7, n=6, ctx=None):
        self.m, self.n, self.ctx = m, n, ctx

    def compute(self, x, out=None):
        return _op(self.ctx, _lib.lib().lele_hip_lfr, [x], [C.c_int64(self.m), C.c_int64(self.n)], out)


class Cmvn:  # cmvn.rs
    def __init__(self, eps=1e-5, ctx=None):
        self.eps, self.ctx = eps, ctx

    def compute(self, x, out=None):
        return _op(self.ctx, _lib.lib().lele_hip_cmvn, [x], [C.c_float(self.eps)], out)

    def apply_with_stats(self, x, mean, std, out=None):
        ctx = _ctx(self.ctx)
        keep = []
        out = out or ctx.buf()
        sh = _lib.OutShape()
        _lib.check(_lib.lib().lele_hip_cmvn_apply_with_stats(
            ctx._h, _lib.as_tensor(unwrap(x), keep), _lib.as_tensor(unwrap(mean), keep),
            _lib.as_tensor(unwrap(std), keep), C.c_float(self.eps), out._h, sh.shape, C.byref(sh.rank)))
        return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


class RealFft:  # features/fft.rs:1-49
    def __init__(self, length, ctx=None):
        self.n, self.ctx = length, ctx

    def process(self, x):
        """x [rows, n] or [n] real -> (re, im) each [rows, n/2+1]"""
        ctx = _ctx(self.ctx)
        keep = []
        a = np.asarray(unwrap(x), np.float32) if not isinstance(unwrap(x), _lib.DevTensor) else unwrap(x)
        if isinstance(a, np.ndarray) and a.ndim == 1:
            a = a.reshape(1, -1)
        o_re, o_im = ctx.buf(), ctx.buf()
        sh = _lib.OutShape()
        _lib.check(_lib.lib().lele_hip_rfft(ctx._h, _lib.as_tensor(a, keep), o_re._h, o_im._h, sh.shape,
                                            C.byref(sh.rank)))
        return (TensorView(_lib.DevTensor(o_re, sh.get(), np.float32)),
                TensorView(_lib.DevTensor(o_im, sh.get(), np.float32)))
