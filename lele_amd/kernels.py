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


def _call(fn, tensors, extra=(), out=None, ctx=None, dtype=np.float32):
    """generic C-ABI call: fn(ctx, *tensors, *extra, out, out_shape, out_rank) -> TensorView"""
    return _op(ctx, fn, tensors, list(extra), out, dtype)


def matmul(a, b, out=None, ctx=None):  # gemm.rs:112
    return _call(_lib.lib().lele_hip_matmul, [a, b], (), out, ctx)


def matmul_fused_add(a, b, bias, out=None, ctx=None):  # gemm.rs:223
    return _call(_lib.lib().lele_hip_matmul_fused_add, [a, b, bias], (), out, ctx)


def gemm(a, b, c=None, alpha=1.0, beta=1.0, trans_a=False, trans_b=False, out=None, ctx=None):  # gemm.rs:433
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_gemm(ctx._h, _lib.as_tensor(unwrap(a), keep), _lib.as_tensor(unwrap(b), keep),
                                        _lib.as_tensor(unwrap(c), keep), C.c_float(alpha), C.c_float(beta),
                                        C.c_int(int(trans_a)), C.c_int(int(trans_b)), out._h, sh.shape,
                                        C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def _mem_weight(x):
    """mark a host array as an immutable weight (LELE_MEM_WEIGHT): uploaded / pre-packed once per ctx"""
    return Weight(x)


class Weight:
    """A host array that is immutable for the life of the ctx (a weights.bin slice)."""

    def __init__(self, arr):
        self.arr = np.ascontiguousarray(np.asarray(arr, dtype=np.float32) if np.asarray(arr).dtype not in
                                        (np.float32, np.int64, np.int32, np.uint8, np.int8) else arr)


def _t(x, keep):
    if isinstance(x, Weight):
        return _lib.as_tensor(x.arr, keep, _lib.MEM_WEIGHT)
    return _lib.as_tensor(unwrap(x), keep)


def fused_quantized_linear(input, weight_int8, weight_scale, weight_zero, bias, apply_relu=False, out=None,
                           ctx=None):  # quantization.rs:77
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fused_quantized_linear(
        ctx._h, _t(input, keep), _t(weight_int8, keep), _t(weight_scale, keep), _t(weight_zero, keep), _t(bias, keep),
        C.c_int(int(apply_relu)), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def dynamic_quantize_linear(x, ctx=None):  # quantization.rs:1628 -> (y, scale, zero_point)
    ctx = _ctx(ctx)
    keep = []
    oy, os_, oz = ctx.buf(), ctx.buf(), ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_dynamic_quantize_linear(ctx._h, _t(x, keep), oy._h, os_._h, oz._h, sh.shape,
                                                           C.byref(sh.rank)))
    return (TensorView(_lib.DevTensor(oy, sh.get(), np.float32)), TensorView(_lib.DevTensor(os_, (1,), np.float32)),
            TensorView(_lib.DevTensor(oz, (1,), np.float32)))


def mat_mul_integer_with_scale_bias(a, b, a_zero_point=None, b_zero_point=None, scale=None, bias=None,
                                    apply_relu=False, out=None, ctx=None):  # quantization.rs:31, 927
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_mat_mul_integer_with_scale_bias(
        ctx._h, _t(a, keep), _t(b, keep), _t(a_zero_point, keep), _t(b_zero_point, keep), _t(scale, keep),
        _t(bias, keep), C.c_int(int(apply_relu)), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def mat_mul_integer(a, b, a_zero_point=None, b_zero_point=None, out=None, ctx=None):  # quantization.rs:8
    return mat_mul_integer_with_scale_bias(a, b, a_zero_point, b_zero_point, None, None, False, out, ctx)


def mat_mul_integer_with_bias(a, b, a_zero_point=None, b_zero_point=None, bias=None, out=None, ctx=None):
    return mat_mul_integer_with_scale_bias(a, b, a_zero_point, b_zero_point, None, bias, False, out, ctx)


def mat_mul_integer_with_scale_bias_relu(a, b, a_zero_point=None, b_zero_point=None, scale=None, bias=None, out=None,
                                         ctx=None):
    return mat_mul_integer_with_scale_bias(a, b, a_zero_point, b_zero_point, scale, bias, True, out, ctx)


# ------------------------------------------------------------------------------------------- element-wise
_UNARY = {"exp": 0, "sigmoid": 1, "tanh_kernel": 2, "silu": 3, "erf": 4, "gelu": 5, "fast_gelu": 6, "relu": 7,
          "sqrt": 8, "log": 9, "sin": 10, "cos": 11, "neg": 12, "reciprocal": 13, "softplus": 14, "not_": 15,
          "abs": 16, "floor": 17, "ceil": 18}
_BINARY = {"add": 0, "sub": 1, "mul": 2, "div": 3, "pow": 4, "max": 5, "min": 6, "equal": 7, "less": 8, "greater": 9,
           "prelu": 10, "mod_f32": 11, "and_": 12, "or_": 13}


def _make_unary(name, op):
    def fn(input, out=None, ctx=None):
        return _op(ctx, _lib.lib().lele_hip_unary, [input], [], out, prefix=[C.c_int(op)])
    fn.__name__ = name
    fn.__doc__ = "lele::kernels::%s (src/kernels/math.rs)" % name.rstrip("_")
    return fn


def _make_binary(name, op):
    def fn(a, b, out=None, ctx=None):
        dt = np.int64 if (np.asarray(unwrap(a)).dtype == np.int64 if not isinstance(unwrap(a), _lib.DevTensor)
                          else unwrap(a).dtype == np.int64) else np.float32
        return _op(ctx, _lib.lib().lele_hip_binary, [a, b], [], out, dt, prefix=[C.c_int(op)])
    fn.__name__ = name
    fn.__doc__ = "lele::kernels::%s (src/kernels/math.rs), numpy-style broadcast" % name.rstrip("_")
    return fn


for _n, _o in _UNARY.items():
    globals()[_n] = _make_unary(_n, _o)
for _n, _o in _BINARY.items():
    globals()[_n] = _make_binary(_n, _o)
tanh = tanh_kernel  # noqa: F821


def where_op(cond, x, y, out=None, ctx=None):  # manipulation.rs:1215
    return _op(ctx, _lib.lib().lele_hip_where, [cond, x, y], [], out)


def clip(input, min=None, max=None, out=None, ctx=None):  # math.rs:1984: min/max are 1-element tensors or None
    lo = None if min is None else float(np.asarray(unwrap(min) if not isinstance(unwrap(min), _lib.DevTensor)
                                                   else TensorView(unwrap(min)).numpy()).reshape(-1)[0])
    hi = None if max is None else float(np.asarray(unwrap(max) if not isinstance(unwrap(max), _lib.DevTensor)
                                                   else TensorView(unwrap(max)).numpy()).reshape(-1)[0])
    return _op(ctx, _lib.lib().lele_hip_clip, [input],
               [C.c_int(lo is not None), C.c_float(lo or 0.0), C.c_int(hi is not None), C.c_float(hi or 0.0)], out)


def _reduce(op, input, axes, keepdims, out, ctx):
    keep = []
    arr, n = _lib.i64_array(list(axes), keep)
    return _op(ctx, _lib.lib().lele_hip_reduce, [input], [arr, n, C.c_int(int(keepdims))], out, prefix=[C.c_int(op)])


def reduce_sum(input, axes, keepdims=True, out=None, ctx=None):  # math.rs:1611
    return _reduce(0, input, axes, keepdims, out, ctx)


def reduce_mean(input, axes, keepdims=True, out=None, ctx=None):  # math.rs:1527
    return _reduce(1, input, axes, keepdims, out, ctx)


def reduce_max(input, axes, keepdims=True, out=None, ctx=None):  # math.rs:1688
    return _reduce(2, input, axes, keepdims, out, ctx)


def reduce_l2(input, axes, keepdims=True, out=None, ctx=None):  # math.rs:1771
    return _reduce(3, input, axes, keepdims, out, ctx)


def layer_norm(input, scale, bias, axis, epsilon, out=None, ctx=None):  # norm.rs:226
    return _op(ctx, _lib.lib().lele_hip_layer_norm, [input, scale, bias], [C.c_int32(axis), C.c_float(epsilon)], out)


def rms_norm(input, weight, axis, epsilon, out=None, ctx=None):  # norm.rs:420
    return _op(ctx, _lib.lib().lele_hip_rms_norm, [input, weight], [C.c_int32(axis), C.c_float(epsilon)], out)


def softmax(input, axis, out=None, ctx=None):  # norm.rs:8
    return _op(ctx, _lib.lib().lele_hip_softmax, [input], [C.c_int32(axis)], out)


def batch_norm(input, scale, bias, mean, var, epsilon, out=None, ctx=None):  # norm.rs:313
    return _op(ctx, _lib.lib().lele_hip_batch_norm, [input, scale, bias, mean, var], [C.c_float(epsilon)], out)
