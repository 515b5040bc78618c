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


SANITIZE = os.environ.get("ORACLE_SANITIZE", "0") not in ("", "0")   # the ASan / UBSan build (needs libasan preloaded: see
                                                                      # tests/test_oracle_sanitized.py, which sets that up)


def build(force=False):
    name = "liboracle_asan.so" if SANITIZE else "liboracle.so"
    so = os.path.join(_HERE, name)
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", name])
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


def scalar_mode():
    return bool(lib().orc_scalar_mode())


class scalar:
    """with scalar(): the operators that have a cfg(not(any(x86_64, aarch64, wasm32))) body upstream run its restatement (oracle/scalar.cpp):
    softmax, layer_norm, the LSTM / GRU gate stages, the unary activations (libm), one-channel conv1d, dynamic_quantize_linear"""

    def __enter__(self):
        self.was = lib().orc_scalar_mode()
        lib().orc_set_scalar_mode(1)
        return self

    def __exit__(self, *exc):
        lib().orc_set_scalar_mode(self.was)
        return False


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


# ---------------------------------------------------------------- gemm
def _batch_dims(shape):
    b = 1
    for d in shape[:-2]:
        b *= d
    return b


def matmul(a, b, acc32=False):
    a, b = _f32(a), _f32(b)
    m, k = a.shape[-2:]
    k2, n = b.shape[-2:]
    assert k == k2, "MatMul K dim mismatch"
    ba, bb = _batch_dims(a.shape), _batch_dims(b.shape)
    assert bb == 1 or bb == ba, "MatMul broadcast not fully supported yet"
    lead = a.shape[:-2] if ba >= bb else b.shape[:-2]
    out = np.empty(tuple(lead) + (m, n), np.float32)
    lib().orc_matmul(_p(a), _p(b), i64(ba), i64(bb), i64(m), i64(k), i64(n), _p(out), C.c_int(int(acc32)))
    return out


def matmul_fused_add(a, b, bias, acc32=False):
    a, b, bias = _f32(a), _f32(b), _f32(bias)
    m, k = a.shape[-2:]
    n = b.shape[-1]
    ba, bb = max(_batch_dims(a.shape), 1), max(_batch_dims(b.shape), 1)
    lead = a.shape[:-2] if ba >= bb else b.shape[:-2]
    out = np.empty(tuple(lead) + (m, n), np.float32)
    lib().orc_matmul_fused_add(_p(a), _p(b), _p(bias), i64(bias.size), i64(ba), i64(bb), i64(m), i64(k), i64(n),
                               _p(out), C.c_int(int(acc32)))
    return out


def gemm(a, b, c=None, alpha=1.0, beta=1.0, trans_a=False, trans_b=False, acc32=False):
    a, b = _f32(a), _f32(b)
    m = a.shape[-1] if trans_a else a.shape[-2]
    k = a.shape[-2] if trans_a else a.shape[-1]
    n = b.shape[-2] if trans_b else b.shape[-1]
    out = np.empty((m, n), np.float32)
    cc = _f32(c) if c is not None else None
    lib().orc_gemm(_p(a), _p(b), _p(cc) if cc is not None else None, i64(cc.size if cc is not None else 0), f(alpha),
                   f(beta), C.c_int(int(trans_a)), C.c_int(int(trans_b)), i64(m), i64(k), i64(n), _p(out),
                   C.c_int(int(acc32)))
    return out


# ---------------------------------------------------------------- quantization
def dynamic_quantize_linear(x):
    x = _f32(x)
    y = np.empty_like(x)
    s, z = C.c_float(), C.c_float()
    lib().orc_dynamic_quantize_linear(_p(x), i64(x.size), _p(y), C.byref(s), C.byref(z))
    return y, np.array([s.value], np.float32), np.array([z.value], np.float32)


def fused_quantized_linear(x, weight, weight_scale, weight_zero, bias, relu=False):
    x, weight, weight_scale = _f32(x), _f32(weight), _f32(weight_scale).reshape(-1)
    m, k = x.shape[-2:]
    n = weight.shape[-1]
    batch = _batch_dims(x.shape)
    wz = float(np.asarray(weight_zero).reshape(-1)[0]) if np.asarray(weight_zero).size else 0.0
    b = _f32(bias).reshape(-1) if bias is not None and np.asarray(bias).size else None
    out = np.empty(x.shape[:-1] + (n,), np.float32)
    lib().orc_fused_quantized_linear(_p(x), i64(batch), i64(m), i64(k), i64(n), _p(weight), _p(weight_scale),
                                     i64(weight_scale.size), f(wz), _p(b) if b is not None else None,
                                     C.c_int(int(relu)), _p(out))
    return out


def mat_mul_integer(a, b, a_zero_point=None, b_zero_point=None, scale=None, bias=None, relu=False):
    a, b = _f32(a), _f32(b)
    m, k = a.shape[-2:]
    n = b.shape[-1]
    ba, bb = _batch_dims(a.shape), _batch_dims(b.shape)
    lead = a.shape[:-2] if ba >= bb else b.shape[:-2]
    out = np.empty(tuple(lead) + (m, n), np.float32)
    za = float(np.asarray(a_zero_point).reshape(-1)[0]) if a_zero_point is not None else 0.0
    zb = float(np.asarray(b_zero_point).reshape(-1)[0]) if b_zero_point is not None else 0.0
    sc = _f32(scale).reshape(-1) if scale is not None else None
    bi = _f32(bias).reshape(-1) if bias is not None else None
    lib().orc_mat_mul_integer(_p(a), _p(b), i64(ba), i64(bb), i64(m), i64(k), i64(n), f(za), f(zb),
                              _p(sc) if sc is not None else None, i64(sc.size if sc is not None else 0),
                              _p(bi) if bi is not None else None, C.c_int(int(relu)), _p(out))
    return out


# ---------------------------------------------------------------- activations / normalisation (SIMD restatement)
UNARY_SIMD = {"exp": 0, "sigmoid": 1, "tanh": 2, "silu": 3, "erf": 4, "gelu": 5, "fast_gelu": 6, "relu": 7, "sqrt": 8}


def unary(name, x):
    x = _f32(x)
    out = np.empty_like(x)
    lib().orc_unary_simd(C.c_int(UNARY_SIMD[name]), _p(x), _p(out), i64(x.size))
    return out


def layer_norm(x, scale, bias, axis=-1, eps=1e-5):
    x, scale, bias = _f32(x), _f32(scale), _f32(bias)
    axis = axis + x.ndim if axis < 0 else axis
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    norm = int(np.prod(x.shape[axis:], dtype=np.int64))
    out = np.empty_like(x)
    lib().orc_layer_norm(_p(x), _p(scale), _p(bias), _p(out), i64(norm), i64(outer), f(eps))
    return out


def softmax(x, axis=-1):
    x = _f32(x)
    axis = axis + x.ndim if axis < 0 else axis
    assert axis == x.ndim - 1 or int(np.prod(x.shape[axis + 1:])) == 1, "Softmax only supported on last dimension"
    n = x.shape[axis]
    out = np.empty_like(x)
    lib().orc_softmax_lastdim(_p(x), _p(out), i64(x.size // max(n, 1)), i64(n))
    return out


def rms_norm(x, weight, axis=-1, eps=1e-5):
    x, weight = _f32(x), _f32(weight)
    axis = axis + x.ndim if axis < 0 else axis
    outer = int(np.prod(x.shape[:axis], dtype=np.int64))
    norm = int(np.prod(x.shape[axis:], dtype=np.int64))
    out = np.empty_like(x)
    lib().orc_rms_norm(_p(x), _p(weight), _p(out), i64(norm), i64(outer), f(eps))
    return out


def batch_norm(x, scale, bias, mean, var, eps=1e-5):
    x = _f32(x)
    shape = x.shape
    c = shape[1] if len(shape) > 1 else shape[0]
    outer = shape[0]
    inner = int(np.prod(shape[2:], dtype=np.int64)) if len(shape) > 2 else 1
    out = np.empty_like(x)
    lib().orc_batch_norm(_p(x), _p(_f32(scale)), _p(_f32(bias)), _p(_f32(mean)), _p(_f32(var)), f(eps), i64(outer),
                         i64(c), i64(inner), _p(out))
    return out


# ---------------------------------------------------------------- rnn / conv
def lstm(x, w, r, bias=None, h0=None, c0=None):
    x, w, r = _f32(x), _f32(w), _f32(r)
    t, b, i = x.shape
    assert w.shape[0] == 1 and b == 1, "LSTM: Only num_directions=1 / batch_size=1 supported"
    h = w.shape[1] // 4
    y, oh, oc = np.empty((t, 1, 1, h), np.float32), np.empty((1, 1, h), np.float32), np.empty((1, 1, h), np.float32)
    bb = _f32(bias).reshape(-1) if bias is not None else None
    hh = _f32(h0).reshape(-1) if h0 is not None else None
    cc = _f32(c0).reshape(-1) if c0 is not None else None
    lib().orc_lstm(_p(x), i64(t), i64(i), i64(h), _p(w), _p(r), _p(bb) if bb is not None else None,
                   _p(hh) if hh is not None else None, _p(cc) if cc is not None else None, _p(y), _p(oh), _p(oc))
    return y, oh, oc


def gru(x, w, r, bias=None, h0=None):
    x, w, r = _f32(x), _f32(w), _f32(r)
    t, b, i = x.shape
    assert w.shape[0] == 1 and b == 1
    h = w.shape[1] // 3
    y, oh = np.empty((t, 1, 1, h), np.float32), np.empty((1, 1, h), np.float32)
    bb = _f32(bias).reshape(-1) if bias is not None else None
    hh = _f32(h0).reshape(-1) if h0 is not None else None
    lib().orc_gru(_p(x), i64(t), i64(i), i64(h), _p(w), _p(r), _p(bb) if bb is not None else None,
                  _p(hh) if hh is not None else None, _p(y), _p(oh))
    return y, oh


def _attr2(v, default):
    v = list(v)
    if len(v) >= 2:
        return int(v[0]), int(v[1])
    if len(v) == 1:
        return int(v[0]), int(v[0])
    return default, default


def _pads4(p):  # conv2d.rs:246-273: [top, left, bottom, right]; 2 values = symmetric
    p = list(p)
    if len(p) >= 4:
        return int(p[0]), int(p[1]), int(p[2]), int(p[3])
    if len(p) >= 2:
        return int(p[0]), int(p[1]), int(p[0]), int(p[1])
    return 0, 0, 0, 0


ACT = {None: 0, "none": 0, "relu": 1, "silu": 2}


def conv2d(x, w, bias=None, dilations=(), group=1, pads=(), strides=(), act=None):
    x, w = _f32(x), _f32(w)
    n, c, ih, iw = x.shape
    oc, _, kh, kw = w.shape
    dh, dw = _attr2(dilations, 1)
    sh, sw = _attr2(strides, 1)
    pt, pl, pb, pr = _pads4(pads)
    oh = (ih + pt + pb - dh * (kh - 1) - 1) // sh + 1
    ow = (iw + pl + pr - dw * (kw - 1) - 1) // sw + 1
    out = np.empty((n, oc, oh, ow), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_conv2d(_p(x), _p(w), _p(b) if b is not None else None, i64(n), i64(c), i64(ih), i64(iw), i64(oc), i64(kh),
                     i64(kw), i64(group), i64(pt), i64(pl), i64(pb), i64(pr), i64(sh), i64(sw), i64(dh), i64(dw),
                     C.c_int(ACT[act]), _p(out))
    return out


def conv_integer(x, w, x_zp=0.0, w_zp=0.0, dilations=(), group=1, pads=(), strides=()):
    """conv2d_with_zero_points (conv2d.rs:1507-1997) as conv_integer / conv_integer_from_f32 reach it: both operands centred in f32,
    then lele's f32 convolution.  The PADDING is raw zeros: im2col_with_zp writes `neg_zp = -x_zp` into every padded cell
    (conv2d.rs:2025, 2043-2050, 2183-2199 -- "pad value is (0 - x_zp)"), i.e. a padded cell stands for the u8 value 0, not for the
    zero point.  Restated as: pad the u8 image with zeros, centre, convolve without padding."""
    x, w = _f32(x), _f32(w)
    pt, pl, pb, pr = _pads4(pads)
    xp = np.pad(x, ((0, 0), (0, 0), (pt, pb), (pl, pr))) if (pt or pl or pb or pr) else x
    return conv2d(xp - np.float32(x_zp), w - np.float32(w_zp), None, dilations, group, [0, 0, 0, 0], strides)


def conv2d_im2col(x, w, bias=None, dilations=(), group=1, pads=(), strides=(), act=None):
    """conv2d as lele's x86 build runs it (conv_fast.cpp: im2col + k-ordered f32 FMA GEMM + bias / activation pass)"""
    x, w = _f32(x), _f32(w)
    n, c, ih, iw = x.shape
    oc, _, kh, kw = w.shape
    dh, dw = _attr2(dilations, 1)
    sh, sw = _attr2(strides, 1)
    pt, pl, pb, pr = _pads4(pads)
    oh = (ih + pt + pb - dh * (kh - 1) - 1) // sh + 1
    ow = (iw + pl + pr - dw * (kw - 1) - 1) // sw + 1
    out = np.empty((n, oc, oh, ow), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_conv2d_im2col(_p(x), _p(w), _p(b) if b is not None else None, i64(n), i64(c), i64(ih), i64(iw), i64(oc), i64(kh),
                            i64(kw), i64(group), i64(pt), i64(pl), i64(pb), i64(pr), i64(sh), i64(sw), i64(dh), i64(dw),
                            C.c_int(ACT[act]), _p(out))
    return out


def conv1d(x, w, bias=None, dilations=(), group=1, pads=(), strides=(), relu=False):
    """conv1d_fused, conv1d.rs:853-1464: NCL; pads = [left, right]"""
    x, w = _f32(x), _f32(w)
    p = list(pads)
    pl, pr = (int(p[0]), int(p[1])) if len(p) >= 2 else ((int(p[0]), 0) if len(p) == 1 else (0, 0))  # conv1d.rs:886-887
    if x.ndim == 2:  # [N, L] is one channel, conv1d.rs:869
        x = x[:, None, :]
    d = int(list(dilations)[0]) if len(list(dilations)) else 1
    s = int(list(strides)[0]) if len(list(strides)) else 1
    if scalar_mode() and x.shape[1] == 1 and group == 1 and pl == 0 and pr == 0 and d == 1:   # conv1d.rs:902, 945-961 -> 1578-1615
        n, _c, length = x.shape
        oc, _ci, k = w.shape
        ol = (length - k) // s + 1
        out = np.empty((n, oc, ol), np.float32)
        b = _f32(bias) if bias is not None else None
        lib().orc_scalar_conv1d_single_channel(_p(x), _p(w), _p(b) if b is not None else None, i64(n), i64(length), i64(oc), i64(k), i64(s),
                                               i64(ol), C.c_int(1 if relu else 0), _p(out))
        return out
    y = conv2d(x[:, :, None, :], w[:, :, None, :], bias, [1, d], group, [0, pl, 0, pr], [1, s], "relu" if relu else None)
    return y[:, :, 0, :]


def conv_transpose(x, w, bias=None, dilations=(), group=1, pads=(), strides=()):
    assert group == 1, "ConvTranspose: group > 1 not supported yet"
    x, w = _f32(x), _f32(w)
    n, c, ih, iw = x.shape
    _, oc, kh, kw = w.shape
    st, dl, pd = list(strides), list(dilations), list(pads)
    sh = int(st[0]) if len(st) > 0 else 1
    sw = int(st[1]) if len(st) > 1 else 1
    dh = int(dl[0]) if len(dl) > 0 else 1
    dw = int(dl[1]) if len(dl) > 1 else 1
    pt = int(pd[0]) if len(pd) > 0 else 0
    pl = int(pd[1]) if len(pd) > 1 else 0
    pb = int(pd[2]) if len(pd) > 2 else pt
    pr = int(pd[3]) if len(pd) > 3 else pl
    oh = (ih - 1) * sh - (pt + pb) + dh * (kh - 1) + 1
    ow = (iw - 1) * sw - (pl + pr) + dw * (kw - 1) + 1
    out = np.empty((n, oc, oh, ow), np.float32)
    b = _f32(bias) if bias is not None else None
    lib().orc_conv_transpose2d(_p(x), _p(w), _p(b) if b is not None else None, i64(n), i64(c), i64(ih), i64(iw), i64(oc),
                               i64(kh), i64(kw), i64(pt), i64(pl), i64(pb), i64(pr), i64(sh), i64(sw), i64(dh), i64(dw),
                               _p(out))
    return out


# ---------------------------------------------------------------- application-side steps (apps.cpp)
def decode_greedy_ids(logits, skip):
    """tokenizer.rs:50-71 -> (ids [B, T] padded with -1, counts [B])"""
    logits = _f32(logits)
    b, t, v = logits.shape
    skip = np.ascontiguousarray(skip, np.uint8)
    out, counts = np.empty((b, t), np.int32), np.empty(b, np.int32)
    lib().orc_decode_greedy_ids(_p(logits), i64(b), i64(t), i64(v), _p(skip), i64(skip.size), _p(out), _p(counts))
    return out, counts


def image_preprocess(rgb, target=640):
    """yolo26n-seg image.rs:62-111"""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    out = np.empty((1, 3, target, target), np.float32)
    lib().orc_image_preprocess(_p(rgb), i64(rgb.shape[0]), i64(rgb.shape[1]), i64(target), _p(out))
    return out


def yolo_seg_postprocess(logits, mask_features, img_width, img_height, threshold, num_classes=80):
    """yolo26n-seg image.rs:127-265 -> (dets [n, 38], mask u8 [H, W])"""
    logits, mask_features = _f32(logits), _f32(mask_features)
    dets, mask = np.zeros((300, 38), np.float32), np.empty((img_height, img_width), np.uint8)
    lib().orc_yolo_seg_postprocess.restype = C.c_int32
    n = lib().orc_yolo_seg_postprocess(_p(logits), _p(mask_features), i64(mask_features.size), i64(img_width), i64(img_height),
                                       f(threshold), i64(num_classes), _p(dets), _p(mask))
    return dets[:n], mask


def vad_segments(probs, chunk_size, padded_len, audio_len, sample_rate=16000, threshold=0.3, min_silence_ms=200.0,
                 min_speech_ms=400.0, speech_pad_ms=120.0, merge_gap_ms=200.0):
    """silero main.rs:151-228 (defaults: VadConfig::default, main.rs:18-28) -> list of (start, end) sample indices"""
    probs = _f32(probs)
    seg = np.zeros((max(1, probs.size), 2), np.int64)
    lib().orc_vad_segments.restype = C.c_int64
    n = lib().orc_vad_segments(_p(probs), i64(probs.size), i64(chunk_size), i64(padded_len), i64(audio_len), C.c_uint32(sample_rate),
                               f(threshold), f(min_silence_ms), f(min_speech_ms), f(speech_pad_ms), f(merge_gap_ms), _p(seg),
                               i64(seg.shape[0]))
    return [(int(a), int(b)) for a, b in seg[:n]]
