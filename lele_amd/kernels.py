"""lele::kernels on the device (host mirror of /root/reference/src/kernels/mod.rs:23-39 re-exports).

Each function keeps the reference's name and argument order; `out` (a _lib.Buf) plays the role of the
`out: &mut Vec<f32>` workspace buffer and may be omitted.
"""
import builtins as _b
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


def _pitch(x):
    x = unwrap(x)
    return x.pitch if isinstance(x, _lib.DevTensor) else 0


def _is_view(x):
    x = unwrap(x)
    return isinstance(x, _lib.DevTensor) and x.is_view


def _op_pitched(ctx, fn, tensors, extra, out, out_window, dtype=np.float32, prefix=(), pitch_of=(0, 1)):
    """a *_pitched entry point (include/lele_hip.h, LelePitch): fn(ctx, *prefix, *tensors, *extra, pitch, out, out_shape, out_rank).
    The first two tensors may be channel views; out_window = (offset, pitch) in elements writes the result into a window of `out`
    (which must already hold the enclosing tensor) and returns the view of it."""
    ctx = _ctx(ctx)
    keep = []
    args = [ctx._h] + list(prefix)
    for t in tensors:
        args.append(_lib.as_tensor(unwrap(t), keep, views=True))
    for k, t in enumerate(tensors):
        if k not in pitch_of and _is_view(t):
            raise _lib.LeleError("only the operands named by x_pitch / y_pitch may be channel views")
    px, py = (tensors[k] if k < len(tensors) else None for k in pitch_of)
    pv = _lib.LelePitch(_pitch(px) if px is not None else 0, _pitch(py) if py is not None else 0,
                        int(out_window[0]) if out_window else 0, int(out_window[1]) if out_window else 0)
    args.extend(extra)
    out = out or ctx.buf()
    sh = _lib.OutShape()
    args.extend([C.byref(pv), out._h, sh.shape, C.byref(sh.rank)])
    _lib.check(fn(*args))
    return TensorView(_lib.DevTensor(out, sh.get(), dtype, pv.out_offset, pv.out_pitch))


def copy_view(input, out=None, out_window=None, ctx=None):
    """copy a tensor or channel view into `out` (dense) or into a window of it: Concat / Split along C when an operand cannot be
    produced or consumed in place (manipulation.rs:108-207, 1091-1151)"""
    return _op_pitched(ctx, _lib.lib().lele_hip_copy_pitched, [input], [], out, out_window, _dtype_of(input))


def transpose_cp(input, out=None, out_window=None, ctx=None):
    """[N, C, ...] (a tensor or a channel view) -> [N, P, C], dense or into rows of an already reserved [N, P_total, C] buffer
    (lele_hip_transpose_cp_pitched): the detection tail's Concat -> Transpose -> Split without the intermediate tensors"""
    return _op_pitched(ctx, _lib.lib().lele_hip_transpose_cp_pitched, [input], [], out, out_window)


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


Weight = _lib.Weight


def _t(x, keep):
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


class PreparedWeights:
    """PreparedWeights (quantization.rs:198-215): a u8 weight matrix packed once for the i8 matrix cores"""

    def __init__(self, ctx, h, k, n):
        import weakref
        self.ctx, self._h, self.k, self.n = ctx, h, k, n
        ctx._graphs.append(weakref.ref(self))  # closed with the ctx (the handle points into it), like its graphs

    def close(self):
        if self._h:
            _lib.lib().lele_hip_prepared_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prepare_weights(b_data, k, n, ctx=None):  # quantization.rs:221: raw u8 [K, N]
    ctx = _ctx(ctx)
    b = np.ascontiguousarray(np.asarray(b_data, np.uint8).reshape(-1))
    if b.size != k * n:
        raise _lib.LeleError("prepare_weights: %d bytes given for k=%d, n=%d" % (b.size, k, n))
    h = C.c_void_p()
    _lib.check(_lib.lib().lele_hip_prepare_weights(ctx._h, b.ctypes.data_as(C.c_void_p), C.c_int64(k), C.c_int64(n), C.byref(h)))
    return PreparedWeights(ctx, h, k, n)


def mat_mul_integer_prepared(a, pw, a_zero_point=None, b_zero_point=None, scale=None, bias=None, apply_relu=False, out=None,
                             ctx=None):  # quantization.rs:699; zero points are host scalars (None = Option::None)
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_mat_mul_integer_prepared(
        ctx._h, _t(a, keep), pw._h, C.c_int(a_zero_point is not None), C.c_float(a_zero_point or 0.0),
        C.c_int(b_zero_point is not None), C.c_int32(int(b_zero_point or 0)), _t(scale, keep), _t(bias, keep),
        C.c_int(int(apply_relu)), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


_U8_PREPARED = {}  # (ctx id, data address, bytes) -> PreparedWeights: the shim-side handle cache of mat_mul_integer_u8_weights


def mat_mul_integer_u8_weights(a, b_u8_data, b_shape, a_zero_point=None, b_zero_point=None, scale=None, bias=None,
                               apply_relu=False, out=None, ctx=None):  # quantization.rs:173
    ctx = _ctx(ctx)
    b = np.ascontiguousarray(np.asarray(b_u8_data, np.uint8))
    key = (id(ctx), b.ctypes.data, b.size)
    pw = _U8_PREPARED.get(key)
    if pw is None or pw._h is None:
        pw = _U8_PREPARED[key] = prepare_weights(b, int(b_shape[-2]), int(b_shape[-1]), ctx=ctx)
        pw._keep = b
    return mat_mul_integer_prepared(a, pw, a_zero_point, b_zero_point, scale, bias, apply_relu, out, ctx)


def fused_dq_gemm_prepared(input, pw, b_zero_point, weight_scale, bias=None, apply_relu=False, out=None, ctx=None):
    """fused_dq_gemm_prepared_x86, quantization.rs:454"""
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fused_dq_gemm_prepared(
        ctx._h, _t(input, keep), pw._h, C.c_int(b_zero_point is not None), C.c_int32(int(b_zero_point or 0)),
        _t(weight_scale, keep), _t(bias, keep), C.c_int(int(apply_relu)), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def dynamic_quantize_linear(x, outs=None, ctx=None):  # quantization.rs:1628 -> (y, scale, zero_point); outs = its 3 buffers
    ctx = _ctx(ctx)
    keep = []
    oy, os_, oz = outs if outs else (ctx.buf(), ctx.buf(), ctx.buf())
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
    def fn(a, b, out=None, ctx=None, out_window=None):
        if out_window or _is_view(a) or _is_view(b):   # channel views: same-shape f32 operands only (the library checks)
            return _op_pitched(ctx, _lib.lib().lele_hip_binary_pitched, [a, b], [], out, out_window, prefix=[C.c_int(op)])
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


# ------------------------------------------------------------------------------------------- data movement
def _shape_of(x):
    if isinstance(x, (TensorView, _lib.Weight)):
        return tuple(x.shape)
    x = unwrap(x)
    return tuple(x.shape) if isinstance(x, _lib.DevTensor) else tuple(np.asarray(x).shape)


def _dtype_of(x):
    x = unwrap(x)
    if isinstance(x, _lib.Weight):
        return x.arr.dtype
    if isinstance(x, _lib.DevTensor):
        return x.dtype
    a = np.asarray(x)
    return a.dtype if a.dtype in (np.float32, np.int64, np.int32) else np.dtype(np.float32)


def _row_major_strides(shape):
    st, acc = [], 1
    for d in reversed(shape):
        st.append(acc)
        acc *= d
    return list(reversed(st))


def _strided(x, oshape, strides, offset=0, mods=None, out=None, ctx=None):
    ctx = _ctx(ctx)
    keep = []
    rank = len(oshape)
    a, _ = _lib.i64_array(list(oshape), keep)
    s, _ = _lib.i64_array(list(strides), keep)
    m = _lib.i64_array(list(mods), keep)[0] if mods is not None else None
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_strided_copy(ctx._h, _lib.as_tensor(unwrap(x), keep), a, s, m, C.c_int32(rank),
                                                C.c_int64(offset), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), _dtype_of(x)))


def slice(input, starts, ends, axes=(), steps=(), out=None, ctx=None):  # manipulation.rs:209-380
    shape = _shape_of(input)
    ndim = len(shape)
    a_start, a_end, a_step = [0] * ndim, list(shape), [1] * ndim
    for i in _b.range(len(starts)):
        ax = i if not len(axes) else (axes[i] + ndim if axes[i] < 0 else axes[i])
        dim = shape[ax]
        step = steps[i] if i < len(steps) else 1
        s64, e64 = int(starts[i]), int(ends[i])
        e_max, e_min = e64 > (2 ** 63 - 1) // 2, e64 < -(2 ** 63) // 2
        start = dim if s64 > dim else (-dim if s64 < -dim else s64)
        end = dim if e_max else (-dim if e_min else (dim if e64 > dim else (-dim if e64 < -dim else e64)))
        ns = start + dim if start < 0 else start
        if e_max:
            ne = dim if step > 0 else -1
        elif e_min:
            ne = 0 if step > 0 else -1
        else:
            ne = end + dim if end < 0 else end
        if step > 0:
            s, e = _b.min(_b.max(ns, 0), dim), _b.min(_b.max(ne, 0), dim)
        else:
            s, e = _b.min(_b.max(ns, 0), dim - 1), _b.min(_b.max(ne, -1), dim - 1)
        a_start[ax], a_end[ax], a_step[ax] = s, e, step
    oshape = []
    for s, e, st in zip(a_start, a_end, a_step):
        oshape.append(_b.max(0, (e - s + st - 1) // st) if st > 0 else _b.max(0, (s - e + (-st) - 1) // (-st)))
    istr = _row_major_strides(shape)
    off = sum(s * t for s, t in zip(a_start, istr))
    return _strided(input, oshape, [t * st for t, st in zip(istr, a_step)], off, None, out, ctx)


def transpose(input, perm=(), out=None, ctx=None):  # manipulation.rs:644-1080
    shape = _shape_of(input)
    ndim = len(shape)
    perm = list(_b.range(ndim))[::-1] if not len(perm) else [p + ndim if p < 0 else p for p in perm]
    istr = _row_major_strides(shape)
    return _strided(input, [shape[p] for p in perm], [istr[p] for p in perm], 0, None, out, ctx)


def expand(input, shape, out=None, ctx=None):  # math.rs:2168-2247
    ishape = _shape_of(input)
    nd = _b.max(len(ishape), len(shape))
    oshape, strides = [], []
    istr = _row_major_strides(ishape)
    for i in _b.range(nd):
        oi, ot = i - (nd - len(ishape)), i - (nd - len(shape))
        din = ishape[oi] if oi >= 0 else 1
        dt = (shape[ot] or din) if ot >= 0 else 1  # 0 means "use the input dimension"
        if din == dt or dt == 1:
            oshape.append(din)
        elif din == 1:
            oshape.append(dt)
        else:
            raise _lib.LeleError("Expand: incompatible shapes %s -> %s" % (ishape, tuple(shape)))
        strides.append(istr[oi] if (oi >= 0 and din != 1) else 0)
    return _strided(input, oshape, strides, 0, None, out, ctx)


def tile(input, repeats, out=None, ctx=None):  # math.rs:2249-2302
    shape = _shape_of(input)
    if len(repeats) != len(shape):
        raise _lib.LeleError("Tile: repeats length must match input rank")
    return _strided(input, [d * int(r) for d, r in zip(shape, repeats)], _row_major_strides(shape), 0, list(shape), out,
                    ctx)


def split(input, axis, splits, outputs=None, ctx=None):  # manipulation.rs:1091-1151 -> list of TensorView
    shape = _shape_of(input)
    ndim = len(shape)
    ax = axis + ndim if axis < 0 else axis
    if not 0 <= ax < ndim:
        raise _lib.LeleError("Split: axis out of bounds")
    if sum(splits) != shape[ax]:
        raise _lib.LeleError("Split: splits sum mismatch")
    istr = _row_major_strides(shape)
    res, pos = [], 0
    for i, sz in enumerate(splits):
        osh = list(shape)
        osh[ax] = int(sz)
        res.append(_strided(input, osh, istr, pos * istr[ax], None, outputs[i] if outputs else None, ctx))
        pos += int(sz)
    return res


split_owned = split  # manipulation.rs:1153-1213: same values, owned storage


def concat(inputs, axis, out=None, ctx=None):  # manipulation.rs:108-207
    ctx = _ctx(ctx)
    keep = []
    ptrs = [_lib.as_tensor(unwrap(t), keep) for t in inputs]
    arr = (C.POINTER(_lib.LeleTensor) * len(ptrs))(*[C.cast(p, C.POINTER(_lib.LeleTensor)) for p in ptrs])
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_concat(ctx._h, arr, C.c_size_t(len(ptrs)), C.c_int64(axis), out._h, sh.shape,
                                          C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), _dtype_of(inputs[0])))


def pad(input, pads, constant_value=None, mode="constant", out=None, ctx=None):  # manipulation.rs:382-587
    shape = _shape_of(input)
    rank = len(shape)
    raw = [_b.max(0, int(p)) for p in pads]
    if len(raw) < rank * 2:  # covers only the trailing dims (manipulation.rs:397-412)
        half = len(raw) // 2
        missing = rank - half
        full = [0] * (rank * 2)
        for i in _b.range(half):
            full[missing + i] = raw[i]
            full[rank + missing + i] = raw[half + i]
        raw = full
    dt = _dtype_of(input)
    cv = 0
    if constant_value is not None:
        c = np.asarray(TensorView(unwrap(constant_value)).numpy() if isinstance(unwrap(constant_value), _lib.DevTensor)
                       else unwrap(constant_value)).reshape(-1)
        if c.size:
            cv = c[0]
    bits = int(np.array([cv], dt).view(np.uint32 if dt.itemsize == 4 else np.uint64)[0])
    m = {"constant": 0, "edge": 1, "reflect": 2}.get(mode)
    if m is None:
        raise _lib.LeleError("Pad: unknown mode %r" % mode)
    ctx = _ctx(ctx)
    keep = []
    p, _ = _lib.i64_array(raw, keep)
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_pad(ctx._h, _lib.as_tensor(unwrap(input), keep), p, C.c_int32(m), C.c_uint64(bits),
                                       out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), dt))


def gather(data, indices, axis, out=None, ctx=None):  # manipulation.rs:589-641
    return _op(ctx, _lib.lib().lele_hip_gather, [data, indices], [C.c_int64(axis)], out, _dtype_of(data))


def gather_elements(input, indices, axis, out=None, ctx=None):  # conv2d.rs:1438-1502
    return _op(ctx, _lib.lib().lele_hip_gather_elements, [input, indices], [C.c_int64(axis)], out)


def resize_nearest(input, scales=None, sizes=None, coordinate_transform_mode="asymmetric", out=None, ctx=None, out_window=None):
    """conv2d.rs:1261-1382"""
    shape = _shape_of(input)
    if len(shape) != 4:
        raise _lib.LeleError("Resize: expected rank-4 input")
    if sizes is not None:
        if len(sizes) < 4:
            raise _lib.LeleError("Resize: sizes must have at least 4 elements")
        oh, ow = int(sizes[2]), int(sizes[3])
        if oh <= 0 or ow <= 0:
            raise _lib.LeleError("Resize: sizes H and W must be positive")  # conv2d.rs:1310-1313
    elif scales is not None:
        sh_ = float(np.float32(scales[2])) if len(scales) >= 3 else 1.0
        sw_ = float(np.float32(scales[3])) if len(scales) >= 4 else 1.0
        oh, ow = int(shape[2] * sh_), int(shape[3] * sw_)  # (in_h as f64 * sh as f64) as u64
    else:
        raise _lib.LeleError("Resize: either scales or sizes must be provided")
    extra = [C.c_int64(oh), C.c_int64(ow), C.c_int(int(coordinate_transform_mode == "asymmetric"))]
    if out_window or _is_view(input):
        return _op_pitched(ctx, _lib.lib().lele_hip_resize_nearest_pitched, [input], extra, out, out_window)
    return _op(ctx, _lib.lib().lele_hip_resize_nearest, [input], extra, out)


def adaptive_avg_pool1d(input, output_len, out=None, ctx=None):  # pooling.rs:1
    """upstream signature is (input: &[f32], output: &mut [f32], channels, input_len, output_len) on flat slices; here the
    input is [.., L] and the result [.., output_len]"""
    return _op(ctx, _lib.lib().lele_hip_adaptive_avg_pool1d, [input], [C.c_int64(int(output_len))], out)


def max_pool2d(input, kernel_shape, strides=(), pads=(), dilations=(), ceil_mode=False, out=None, ctx=None, out_window=None):
    """conv2d.rs:1051-1254"""
    keep = []
    args = []
    for v in (kernel_shape, strides, pads, dilations):
        a, n = _lib.i64_array(list(v), keep)
        args += [a, n]
    if out_window or _is_view(input):
        return _op_pitched(ctx, _lib.lib().lele_hip_max_pool2d_pitched, [input], args + [C.c_int(int(ceil_mode))], out, out_window)
    return _op(ctx, _lib.lib().lele_hip_max_pool2d, [input], args + [C.c_int(int(ceil_mode))], out)


def topk(input, k, axis=-1, largest=True, sorted=True, out_values=None, out_indices=None, ctx=None):
    """conv2d.rs:1385-1435 -> (values, indices); the two `&mut Vec` outputs of the reference are out_values / out_indices"""
    ctx = _ctx(ctx)
    keep = []
    ov, oi = out_values or ctx.buf(), out_indices or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_topk(ctx._h, _lib.as_tensor(unwrap(input), keep), C.c_int64(int(k)),
                                        C.c_int(int(largest)), ov._h, oi._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(ov, sh.get(), np.float32)), TensorView(_lib.DevTensor(oi, sh.get(), np.float32))


def _first(x, default):
    a = np.asarray(TensorView(unwrap(x)).numpy() if isinstance(unwrap(x), _lib.DevTensor) else unwrap(x)).reshape(-1)
    return a[0] if a.size else default


def range(start, limit, delta, out=None, ctx=None):  # math.rs:2033-2055 (f32)
    s, l, d = (np.float32(_first(v, dflt)) for v, dflt in ((start, 0.0), (limit, 0.0), (delta, 1.0)))
    n = int(_b.max(np.ceil(np.float32(np.float32(l - s) / d)), 0.0)) if d != 0 else 0
    ctx = _ctx(ctx)
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_range_f32(ctx._h, C.c_float(float(s)), C.c_float(float(d)), C.c_int64(n), out._h,
                                             sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def constant_of_shape(input, value, dtype=np.float32, out=None, ctx=None):  # shape.rs:122-135
    shape = [int(v) for v in np.asarray(TensorView(unwrap(input)).numpy() if isinstance(unwrap(input), _lib.DevTensor)
                                        else unwrap(input)).reshape(-1)]
    dt = np.dtype(dtype)
    bits = int(np.array([value], dt).view(np.uint32 if dt.itemsize == 4 else np.uint64)[0])
    ctx = _ctx(ctx)
    keep = []
    a, _ = _lib.i64_array(shape, keep)
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fill(ctx._h, a, C.c_int32(len(shape)), C.c_int32(_lib._NP2DT[dt]), C.c_uint64(bits),
                                        out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), dt))


def cast_to_f32(input, out=None, ctx=None):  # utils.rs:66-83
    return _op(ctx, _lib.lib().lele_hip_cast, [input], [C.c_int32(_lib.F32)], out, np.float32)


def cast_to_i64(input, out=None, ctx=None):  # utils.rs:84-101
    return _op(ctx, _lib.lib().lele_hip_cast, [input], [C.c_int32(_lib.I64)], out, np.int64)


def reinterpret_as_u8(input, out=None, ctx=None):  # tensor.rs:88-98: f32-carried u8 codes -> u8 (Rust's saturating `as u8`)
    return _op(ctx, _lib.lib().lele_hip_cast, [input], [C.c_int32(_lib.U8)], out, np.uint8)

# ------------------------------------------------------------------------------------------------- conv / rnn
def _conv(fn, input, weights, bias, dilations, group, pads, strides, tail, out, ctx, out_window=None):
    keep = []
    args = []
    d, n = _lib.i64_array(list(dilations), keep)
    args += [d, n, C.c_int64(int(group))]
    for v in (pads, strides):
        a, n = _lib.i64_array(list(v), keep)
        args += [a, n]
    if out_window or _is_view(input):
        if fn is not _lib.lib().lele_hip_conv2d:
            raise _lib.LeleError("channel views are supported by conv2d / conv2d_fused / conv2d_silu only")
        return _op_pitched(ctx, _lib.lib().lele_hip_conv2d_pitched, [input, weights, bias], args + list(tail), out, out_window)
    return _op(ctx, fn, [input, weights, bias], args + list(tail), out)


def conv2d_res(input, weights, bias, res, dilations=(), group=1, pads=(), strides=(), act=0, out=None, ctx=None, out_window=None):
    """act(conv2d(input) + bias) + res in one call (lele_hip_conv2d_res): a residual block's conv2d / conv2d_silu / conv2d_fused and
    the `add` behind it (plan.fuse_residual_adds), the same bits.  act: 0 none, 1 ReLU, 2 SiLU.  input, res and the result may be
    channel views."""
    keep = []
    args = []
    d, n = _lib.i64_array(list(dilations), keep)
    args += [d, n, C.c_int64(int(group))]
    for v in (pads, strides):
        a, n = _lib.i64_array(list(v), keep)
        args += [a, n]
    return _op_pitched(ctx, _lib.lib().lele_hip_conv2d_res, [input, weights, bias, res], args + [C.c_int(int(act))], out, out_window,
                       pitch_of=(0, 3))


def reset_conv_stats(ctx=None):  # conv2d.rs:101 (a no-op upstream; here: clears the context's convolution counters)
    _lib.check(_lib.lib().lele_hip_conv_stats_reset(_ctx(ctx)._h))


def conv_stats(ctx=None):
    """(calls, multiply-accumulates) of the 2-D convolutions issued on the context since the last reset_conv_stats()"""
    calls, macs = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.lib().lele_hip_conv_stats(_ctx(ctx)._h, C.byref(calls), C.byref(macs)))
    return calls.value, macs.value


def print_conv_stats(ctx=None):  # conv2d.rs:75
    calls, macs = conv_stats(ctx)
    print("conv stats: %d convolution calls, %.3f GMAC" % (calls, macs * 1e-9))


def conv2d(input, weights, bias=None, dilations=(), group=1, pads=(), strides=(), out=None, ctx=None, out_window=None):  # conv2d.rs:107
    return _conv(_lib.lib().lele_hip_conv2d, input, weights, bias, dilations, group, pads, strides, [C.c_int(0)], out, ctx, out_window)


def conv2d_fused(input, weights, bias=None, dilations=(), group=1, pads=(), strides=(), relu=False, out=None, ctx=None, out_window=None):
    """conv2d.rs:155"""
    return _conv(_lib.lib().lele_hip_conv2d, input, weights, bias, dilations, group, pads, strides,
                 [C.c_int(1 if relu else 0)], out, ctx, out_window)


def conv2d_silu(input, weights, bias=None, dilations=(), group=1, pads=(), strides=(), out=None, ctx=None, out_window=None):  # conv2d.rs:124
    return _conv(_lib.lib().lele_hip_conv2d, input, weights, bias, dilations, group, pads, strides, [C.c_int(2)], out, ctx, out_window)


def conv1d(input, weights, bias=None, dilations=(), group=1, pads=(), strides=(), out=None, ctx=None):  # conv1d.rs:837
    return _conv(_lib.lib().lele_hip_conv1d, input, weights, bias, dilations, group, pads, strides, [C.c_int(0)], out, ctx)


def conv1d_fused(input, weights, bias=None, dilations=(), group=1, pads=(), strides=(), relu=False, out=None, ctx=None):
    """conv1d.rs:853"""
    return _conv(_lib.lib().lele_hip_conv1d, input, weights, bias, dilations, group, pads, strides,
                 [C.c_int(int(bool(relu)))], out, ctx)


def conv_transpose(input, weights, bias=None, dilations=(), group=1, pads=(), strides=(), out=None, ctx=None):
    """conv2d.rs:2952"""
    return _conv(_lib.lib().lele_hip_conv_transpose, input, weights, bias, dilations, group, pads, strides, [], out, ctx)


def lstm(input, w, r, bias=None, sequence_lens=None, initial_h=None, initial_c=None, outs=None, ctx=None):
    """rnn.rs:67 -> (Y [T,1,1,H], H_n [1,1,H], C_n [1,1,H]); outs = the three `&mut Vec` output buffers of the reference"""
    ctx = _ctx(ctx)
    keep = []
    oy, oh, oc = outs if outs else (ctx.buf(), ctx.buf(), ctx.buf())
    sh = _lib.OutShape()
    t = [_lib.as_tensor(unwrap(v), keep) for v in (input, w, r, bias, sequence_lens, initial_h, initial_c)]
    _lib.check(_lib.lib().lele_hip_lstm(ctx._h, *t, oy._h, oh._h, oc._h, sh.shape, C.byref(sh.rank)))
    ys = sh.get()
    hs = (1, 1, ys[-1])
    return (TensorView(_lib.DevTensor(oy, ys, np.float32)), TensorView(_lib.DevTensor(oh, hs, np.float32)),
            TensorView(_lib.DevTensor(oc, hs, np.float32)))


def gru(input, w, r, bias=None, initial_h=None, linear_before_reset=False, outs=None, ctx=None):
    """rnn.rs:246 -> (Y [T,1,1,H], H_n [1,1,H]); outs = the two `&mut Vec` output buffers of the reference"""
    ctx = _ctx(ctx)
    keep = []
    oy, oh = outs if outs else (ctx.buf(), ctx.buf())
    sh = _lib.OutShape()
    t = [_lib.as_tensor(unwrap(v), keep) for v in (input, w, r, bias, initial_h)]
    _lib.check(_lib.lib().lele_hip_gru(ctx._h, *t, C.c_int(int(bool(linear_before_reset))), oy._h, oh._h, sh.shape,
                                       C.byref(sh.rank)))
    ys = sh.get()
    return TensorView(_lib.DevTensor(oy, ys, np.float32)), TensorView(_lib.DevTensor(oh, (1, 1, ys[-1]), np.float32))



# views: no data movement (shape.rs:2-186)
def _view(input, shape):
    x = unwrap(input)
    return TensorView(x, shape) if isinstance(x, _lib.DevTensor) else TensorView(np.asarray(x), shape)


def _try_reshape(ishape, target, total):
    new, known, infer = [], 1, None
    for i, d in enumerate(target):
        if d == -1:
            if infer is not None:
                return None
            infer = i
        elif d == 0:
            if i >= len(ishape):
                return None
            new.append(ishape[i])
            known *= ishape[i]
        else:
            new.append(int(d))
            known *= int(d)
    if infer is not None:
        if known == 0 or total % known:
            return None
        new.insert(infer, total // known)
    return new if int(np.prod(new, dtype=np.int64)) == total else None


def reshape(input, target_shape_raw):  # shape.rs:2-52 (three strategies)
    ishape = _shape_of(input)
    total = int(np.prod(ishape, dtype=np.int64))
    tgt = [int(v) for v in target_shape_raw]
    r = _try_reshape(ishape, tgt, total) or _try_reshape(ishape, [-1 if d == 0 else d for d in tgt], total)
    if r is None and len(tgt) > len(ishape) > 0:
        col = [tgt[0] if tgt[0] > 0 else -1, -1] + [d if d > 0 else -1 for d in tgt[len(tgt) - (len(ishape) - 1):]]
        r = _try_reshape(ishape, col, total)
    if r is None:
        raise _lib.LeleError("Reshape: element count mismatch (input=%s target=%s)" % (list(ishape), tgt))
    return _view(input, r)


def flatten(input, axis):  # shape.rs:105-121
    s = _shape_of(input)
    ax = axis + len(s) if axis < 0 else axis
    return _view(input, [int(np.prod(s[:ax], dtype=np.int64)), int(np.prod(s[ax:], dtype=np.int64))])


def unsqueeze(input, axes):  # shape.rs:136-156
    s = list(_shape_of(input))
    rank = len(s) + len(axes)
    for a in sorted(axes):
        idx = a + rank if a < 0 else a
        s.insert(idx, 1) if idx <= len(s) else s.append(1)
    return _view(input, s)


def squeeze(input, axes=None):  # shape.rs:157-185
    s = _shape_of(input)
    if axes is None:
        return _view(input, [d for d in s if d != 1])
    ax = {a + len(s) if a < 0 else a for a in axes}
    return _view(input, [d for i, d in enumerate(s) if not (d == 1 and i in ax)])


def identity(input):  # shape.rs:186
    return _view(input, _shape_of(input))


def shape(input):  # shape.rs:100-104 -> i64 [rank]
    return TensorView(np.array(_shape_of(input), np.int64))


def size(input):  # shape.rs:95-99 -> i64 scalar
    return TensorView(np.array(int(np.prod(_shape_of(input), dtype=np.int64)), np.int64).reshape(()))


# ------------------------------------------------------------------------------------------- app-side pre/post
def wav_to_f32(payload, bits_per_sample=16, num_channels=1, out=None, ctx=None):
    """examples/sensevoice/src/audio.rs:52-73: WAV payload bytes (uint8) -> f32 mono samples"""
    b = np.frombuffer(payload, np.uint8) if isinstance(payload, (bytes, bytearray, memoryview)) else payload
    return _op(ctx, _lib.lib().lele_hip_wav_to_f32, [b], [C.c_int32(int(bits_per_sample)), C.c_int32(int(num_channels))], out)


def argmax_last(input, out=None, ctx=None):
    """examples/sensevoice/src/tokenizer.rs:50-61: greedy ids, last of equal maxima (Iterator::max_by) -> int32"""
    return _op(ctx, _lib.lib().lele_hip_argmax_last, [input], [], out, np.int32)


def token_filter(ids, skip, out_ids=None, out_counts=None, ctx=None):
    """tokenizer.rs:63-71: kept ids in frame order (skip[id] set for blank / <|...|> specials) -> (ids padded with -1, counts)"""
    ctx = _ctx(ctx)
    keep = []
    oi, oc = out_ids or ctx.buf(), out_counts or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_token_filter(ctx._h, _lib.as_tensor(unwrap(ids), keep),
                                                _lib.as_tensor(unwrap(skip), keep),
                                                oi._h, oc._h, sh.shape, C.byref(sh.rank)))
    shape = sh.get()
    return TensorView(_lib.DevTensor(oi, shape, np.int32)), TensorView(_lib.DevTensor(oc, shape[:-1], np.int32))


def image_preprocess(rgb, target=640, out=None, ctx=None):
    """examples/yolo26n-seg/src/image.rs:62-111: u8 [H, W, 3] -> f32 [1, 3, target, target] (nearest resize, / 255)"""
    return _op(ctx, _lib.lib().lele_hip_image_preprocess, [rgb], [C.c_int32(int(target))], out)


def yolo_seg_postprocess(logits, mask_features, img_width, img_height, threshold, num_classes=80, out_dets=None, out_count=None,
                         out_mask=None, ctx=None):
    """image.rs:127-265, per image of a batch -> (dets f32 [N, 300, 38] of which the first count[n] rows of image n are valid and the
    rest zeros, count i32 [N], mask u8 [N, H, W]); a single image ([1, 300, 38] / [300, 38]) gives [300, 38], [1], [H, W] as before"""
    ctx = _ctx(ctx)
    keep = []
    od, oc, om = out_dets or ctx.buf(), out_count or ctx.buf(), out_mask or ctx.buf()
    n = int(np.prod(unwrap(logits).shape, dtype=np.int64)) // (300 * 38)
    _lib.check(_lib.lib().lele_hip_yolo_seg_postprocess(ctx._h, _lib.as_tensor(unwrap(logits), keep),
                                                        _lib.as_tensor(unwrap(mask_features), keep), C.c_int32(int(img_width)),
                                                        C.c_int32(int(img_height)), C.c_float(float(threshold)),
                                                        C.c_int32(int(num_classes)), od._h, oc._h, om._h))
    lead = [n] if n != 1 else []
    return (TensorView(_lib.DevTensor(od, lead + [300, 38], np.float32)), TensorView(_lib.DevTensor(oc, [n], np.int32)),
            TensorView(_lib.DevTensor(om, lead + [int(img_height), int(img_width)], np.uint8)))


def _reshape_strides(shape, strides, new):
    """strides of `new` over the same memory, or None when the view cannot be reshaped without a copy"""
    old = [(d, st) for d, st in zip(shape, strides) if d != 1]
    out = [0] * len(new)
    oi, ni = 0, 0
    while ni < len(new) and oi < len(old):
        if new[ni] == 1:
            ni += 1
            continue
        np_, op, nj, oj = new[ni], old[oi][0], ni + 1, oi + 1
        while np_ != op:
            if np_ < op:
                if nj >= len(new):
                    return None
                np_ *= new[nj]
                nj += 1
            else:
                if oj >= len(old):
                    return None
                op *= old[oj][0]
                oj += 1
        for k in _b.range(oi, oj - 1):                       # the merged old dims must be contiguous among themselves
            if old[k][1] != old[k + 1][0] * old[k + 1][1]:
                return None
        out[nj - 1] = old[oj - 1][1]
        for k in _b.range(nj - 1, ni, -1):
            out[k - 1] = out[k] * new[k]
        ni, oi = nj, oj
    if oi < len(old) or any(d != 1 for d in new[ni:]):
        return None
    return out


def _walk_chain(shape, chain):
    """(shape, strides, offset) of the view `chain` describes over a contiguous tensor of `shape`"""
    shape = list(shape)
    strides = _row_major_strides(shape)
    offset = 0
    for step in chain:
        if step[0] == "slice":
            _, axis, start, length = step
            axis = axis + len(shape) if axis < 0 else axis
            if start < 0 or start + length > shape[axis]:
                raise _lib.LeleError("view: slice [%d, %d) outside dimension %d" % (start, start + length, shape[axis]))
            offset += start * strides[axis]
            shape[axis] = length
        elif step[0] == "reshape":
            total = int(np.prod(shape)) if shape else 1
            new = _try_reshape(shape, list(step[1]), total)
            st = _reshape_strides(shape, strides, new) if new is not None else None
            if st is None:
                raise _lib.LeleError("view: reshape %s -> %s needs a copy at this point of the chain" % (shape, list(step[1])))
            shape, strides = list(new), st
        elif step[0] == "transpose":
            perm = [p + len(shape) if p < 0 else p for p in step[1]]
            shape, strides = [shape[p] for p in perm], [strides[p] for p in perm]
        else:
            raise _lib.LeleError("view: unknown step %r" % (step[0],))
    return shape, strides, offset


def _materialise_chain(x, chain, out=None, ctx=None):
    """the chain run as the operators it stands for -- slice, reshape (of a contiguous tensor: always a view), transpose -- with
    real copies: the fallback when the chain is not one strided view of its source (a Reshape that would need a copy mid-chain)"""
    ctx = _ctx(ctx)
    cur = x
    steps = list(chain)
    for i, step in enumerate(steps):
        last = i == len(steps) - 1
        if step[0] == "slice":
            _, axis, start, length = step
            cur = slice(cur, [start], [start + length], [axis], [1], out=out if last else ctx.buf(), ctx=ctx)
        elif step[0] == "reshape":
            cur = reshape(cur, list(step[1]))
        elif step[0] == "transpose":
            cur = transpose(cur, list(step[1]), out=out if last else ctx.buf(), ctx=ctx)
        else:
            raise _lib.LeleError("view: unknown step %r" % (step[0],))
    return cur


def view_copy(input, chain, out=None, ctx=None):
    """One strided copy for a chain of views of `input`: ["slice", axis, start, length], ["reshape", dims] (0 copies the
    dimension, one -1 is inferred: shape.rs:2-13), ["transpose", perm].  Equal, bit for bit, to running slice / reshape /
    transpose one after the other (they are exact copies); emitted by lele_amd.compiler for Split -> Reshape -> Transpose.
    A chain that is not ONE strided view of its source (shapes are not known when the plan is compiled) runs step by step."""
    try:
        shape, strides, offset = _walk_chain(_shape_of(input), chain)
    except _lib.LeleError:
        res = _materialise_chain(input, chain, out, ctx)
        if out is not None and (not isinstance(unwrap(res), _lib.DevTensor) or unwrap(res).buf is not out):
            res = _strided(res, list(_shape_of(res)), _row_major_strides(list(_shape_of(res))), 0, None, out, ctx)
        return res
    return _strided(input, shape, strides, offset, None, out, ctx)


class _MatView(C.Structure):
    _fields_ = [("offset", C.c_int64), ("stride_outer", C.c_int64), ("stride_inner", C.c_int64), ("stride_row", C.c_int64),
                ("stride_col", C.c_int64)]


def matmul_view(a, a_chain, b, b_chain, out_perm=None, out_reshape=None, out=None, ctx=None):
    """matmul(view(a), view(b)) with the views never materialised and, optionally, the product stored as
    transpose(result, out_perm) [then reshaped]: bit-identical to view_copy + matmul + transpose (same kernels, same tiles)."""
    try:
        ash, ast, aoff = _walk_chain(_shape_of(a), a_chain or [])
        bsh, bst, boff = _walk_chain(_shape_of(b), b_chain or [])
        direct = len(ash) == len(bsh) and 2 <= len(ash) <= 4 and ash[:-2] == bsh[:-2] and \
            (ast[-1] == 1 or ast[-2] == 1) and (bst[-1] == 1 or bst[-2] == 1) and \
            (out_perm is None or out_perm[-1] % len(out_perm) == len(out_perm) - 1)
    except _lib.LeleError:
        direct = False
    if not direct:
        # geometries the strided GEMM does not take (a view that needs a copy, a rank-2 run-time B against a batched A, operands
        # without a unit stride, a store that is not row-contiguous): the node sequence this op stands for -- materialise the
        # views, `matmul` (which broadcasts an un-batched B, gemm.rs:131), transpose / reshape the product
        ctx = _ctx(ctx)
        am = _materialise_chain(a, a_chain or [], ctx=ctx) if a_chain else a
        bm = _materialise_chain(b, b_chain or [], ctx=ctx) if b_chain else b
        res = matmul(am, bm, out=None if (out_perm or out_reshape is not None) else out, ctx=ctx)
        if out_perm:
            res = transpose(res, list(out_perm), out=out, ctx=ctx)
        if out_reshape is not None:
            res = reshape(res, list(out_reshape))
        return res
    if ash[-1] != bsh[-2]:
        raise _lib.LeleError("MatMul K dim mismatch: %d vs %d" % (ash[-1], bsh[-2]))
    m, k, n = ash[-2], ash[-1], bsh[-1]
    batch = ash[:-2]
    bo, bi = (batch + [1, 1])[:2] if batch else (1, 1)

    def bstr(st):
        lead = st[:-2]
        return (lead[0] if len(lead) >= 1 else 0), (lead[1] if len(lead) == 2 else 0)
    logical = batch + [m, n]
    perm = list(out_perm) if out_perm else list(_b.range(len(logical)))
    phys = [logical[p] for p in perm]
    pstr = _row_major_strides(phys)
    lstr = [pstr[perm.index(j)] for j in _b.range(len(logical))]
    oshape = phys
    if out_reshape is not None:
        oshape = _try_reshape(phys, list(out_reshape), int(np.prod(phys)) if phys else 1)
        if oshape is None:
            raise _lib.LeleError("matmul_view: cannot reshape %s to %s" % (phys, list(out_reshape)))
    av = _MatView(aoff, *bstr(ast), ast[-2], ast[-1])
    bv = _MatView(boff, *bstr(bst), bst[-2], bst[-1])
    ov = _MatView(0, *bstr(lstr), lstr[-2], lstr[-1])
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    dims, _ = _lib.i64_array(oshape, keep)
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_matmul_view(ctx._h, _lib.as_tensor(unwrap(a), keep), C.byref(av), _lib.as_tensor(unwrap(b), keep), C.byref(bv),
                                               C.c_int64(bo), C.c_int64(bi), C.c_int64(m), C.c_int64(k), C.c_int64(n), C.byref(ov), dims,
                                               C.c_int32(len(oshape)), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def attention_view(q, q_chain, k, k_chain, v, v_chain, scale, out_perm=None, out_reshape=None, out=None, ctx=None):
    """matmul_view(q view, k view) -> softmax_scaled(., scale) -> matmul_view(., v view, out_perm, out_reshape) in ONE launch
    (lele_hip_attention_view): q view [.., T, Dh], k view [.., Dh, Tk] (K transposed), v view [.., Tk, Dh]; the score and
    probability tensors never reach HBM.  Geometries the kernel does not take (head dimension != 128, more than 512 keys,
    non-unit inner strides, LELE_HIP_ATTENTION_FUSED=0) run the three-call sequence this op stands for."""
    import os
    ctx = _ctx(ctx)
    try:
        qsh, qst, qoff = _walk_chain(_shape_of(q), q_chain or [])
        ksh, kst, koff = _walk_chain(_shape_of(k), k_chain or [])
        vsh, vst, voff = _walk_chain(_shape_of(v), v_chain or [])
        is_view = True
    except _lib.LeleError:  # a chain that is not ONE strided view of its source (a Reshape that needs a copy mid-chain): the
        is_view = False     # three calls this op stands for materialise it, exactly as the unfused statements did
        qsh = ksh = vsh = qst = kst = vst = [0, 0]
        qoff = koff = voff = 0
    fused = (is_view and os.environ.get("LELE_HIP_ATTENTION_FUSED", "1") != "0" and len(qsh) == len(ksh) == len(vsh) and 2 <= len(qsh) <= 4
             and qsh[:-2] == ksh[:-2] == vsh[:-2] and qsh[-1] == 128 and ksh[-2] == 128 and vsh[-1] == 128 and ksh[-1] == vsh[-2] <= 512
             and qst[-1] == 1 and kst[-2] == 1 and vst[-1] == 1 and all(s % 4 == 0 for s in qst[:-1] + kst[:-2] + kst[-1:]) and qoff % 4 == 0 and koff % 4 == 0
             and (out_perm is None or out_perm[-1] % len(out_perm) == len(out_perm) - 1))
    # the library runs one workgroup per 32 query rows of a head, per 16 rows when that would leave most of the chip idle (a single
    # utterance); grids smaller still (fewer than 96 blocks of 16 rows) run the three-call sequence
    if fused and int(np.prod(qsh[:-2], dtype=np.int64)) * -(-qsh[-2] // 16) < int(os.environ.get("LELE_HIP_ATTENTION_MIN_BLOCKS", "96")):
        fused = False
    if not fused:
        pool = getattr(ctx, "_attn_tmp", None)
        if pool is None:
            pool = ctx._attn_tmp = {}
        tmp = pool.get(ctx.cur_lane)
        if tmp is None:  # scores / probabilities of the sequence: two buffers kept with the ctx, per lane (two lanes may run one each)
            tmp = pool[ctx.cur_lane] = (ctx.buf(), ctx.buf())
        sc = matmul_view(q, q_chain, k, k_chain, out=tmp[0], ctx=ctx)
        pr = softmax_scaled(sc, scale, -1, out=tmp[1], ctx=ctx) if scale is not None else softmax(sc, -1, out=tmp[1], ctx=ctx)
        return matmul_view(pr, [], v, v_chain, out_perm, out_reshape, out=out, ctx=ctx)
    tq, dh, tk = qsh[-2], qsh[-1], ksh[-1]
    batch = qsh[:-2]
    bo, bi = (batch + [1, 1])[:2] if batch else (1, 1)

    def bstr(st):
        lead = st[:-2]
        return (lead[0] if len(lead) >= 1 else 0), (lead[1] if len(lead) == 2 else 0)
    logical = batch + [tq, dh]
    perm = list(out_perm) if out_perm else list(_b.range(len(logical)))
    phys = [logical[p] for p in perm]
    pstr = _row_major_strides(phys)
    lstr = [pstr[perm.index(j)] for j in _b.range(len(logical))]
    oshape = phys
    if out_reshape is not None:
        oshape = _try_reshape(phys, list(out_reshape), int(np.prod(phys)) if phys else 1)
        if oshape is None:
            raise _lib.LeleError("attention_view: cannot reshape %s to %s" % (phys, list(out_reshape)))
    qv = _MatView(qoff, *bstr(qst), qst[-2], qst[-1])
    kv = _MatView(koff, *bstr(kst), kst[-2], kst[-1])
    vv = _MatView(voff, *bstr(vst), vst[-2], vst[-1])
    ov = _MatView(0, *bstr(lstr), lstr[-2], lstr[-1])
    keep = []
    out = out or ctx.buf()
    dims, _ = _lib.i64_array(oshape, keep)
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_attention_view(
        ctx._h, _t(q, keep), C.byref(qv), _t(k, keep), C.byref(kv), _t(v, keep), C.byref(vv), C.c_int64(bo), C.c_int64(bi),
        C.c_int64(tq), C.c_int64(tk), C.c_int64(dh), _t(scale, keep), C.byref(ov), dims, C.c_int32(len(oshape)), out._h, sh.shape,
        C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


# ------------------------------------------------------------------------------------------- fused forms (lele_amd.compiler)
def fused_quantized_linear_residual(input, weight_int8, weight_scale, weight_zero, bias, apply_relu, res1, res2=None, out=None, ctx=None):
    """((fused_quantized_linear(...) + res1) + res2): the Adds that follow a projection, folded into its store"""
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fused_quantized_linear_residual(
        ctx._h, _t(input, keep), _t(weight_int8, keep), _t(weight_scale, keep), _t(weight_zero, keep), _t(bias, keep),
        C.c_int(int(apply_relu)), _t(res1, keep), _t(res2, keep), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def fused_quantized_linear_residual_ln(input, weight_int8, weight_scale, weight_zero, bias, apply_relu, res1, res2, ln_scale, ln_bias,
                                       epsilon, outs=None, ctx=None):
    """(x1, layer_norm(x1, ln_scale, ln_bias, -1, epsilon)) with x1 = fused_quantized_linear[_residual](...): the projection, the Adds
    behind it and the LayerNorm that reads their sum as one call (bit for bit the two calls)"""
    ctx = _ctx(ctx)
    keep = []
    outs = list(outs) if outs else [ctx.buf(), ctx.buf()]
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fused_quantized_linear_residual_ln(
        ctx._h, _t(input, keep), _t(weight_int8, keep), _t(weight_scale, keep), _t(weight_zero, keep), _t(bias, keep),
        C.c_int(int(apply_relu)), _t(res1, keep), _t(res2, keep), _t(ln_scale, keep), _t(ln_bias, keep), C.c_float(float(epsilon)),
        outs[0]._h, outs[1]._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(outs[0], sh.get(), np.float32)), TensorView(_lib.DevTensor(outs[1], sh.get(), np.float32))


def sanm_out_block(input, weight_int8, weight_scale, weight_zero, bias, apply_relu, v_src, fsmn_w, fsmn_bias, x_offset, pad_left, pad_right,
                   res2, ln_scale, ln_bias, epsilon, outs=None, ctx=None):
    """(x1, layer_norm(x1)) with x1 = fused_quantized_linear_residual(input, W.., relu, depthwise_conv1d_tlc(v_src, fsmn_w, fsmn_bias,
    pad_left, pad_right, False, x_offset, add_input=True), res2): the output half of a SAN-M attention block as one call"""
    ctx = _ctx(ctx)
    keep = []
    outs = list(outs) if outs else [ctx.buf(), ctx.buf()]
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_sanm_out_block(
        ctx._h, _t(input, keep), _t(weight_int8, keep), _t(weight_scale, keep), _t(weight_zero, keep), _t(bias, keep),
        C.c_int(int(apply_relu)), _t(v_src, keep), _t(fsmn_w, keep), _t(fsmn_bias, keep), C.c_int64(int(x_offset)), C.c_int64(int(pad_left)),
        C.c_int64(int(pad_right)), _t(res2, keep), _t(ln_scale, keep), _t(ln_bias, keep), C.c_float(float(epsilon)), outs[0]._h, outs[1]._h,
        sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(outs[0], sh.get(), np.float32)), TensorView(_lib.DevTensor(outs[1], sh.get(), np.float32))


def fused_ffn_quantized(input, w1_int8, w1_scale, w1_zero, b1, w2_int8, w2_scale, w2_zero, b2, apply_relu2=False, res1=None, res2=None,
                        out=None, ctx=None):
    """fused_quantized_linear[_residual](fused_quantized_linear(input, w1.., True), w2.., apply_relu2, res1, res2), bit for bit: a
    transformer layer's feed-forward block whose f32 hidden tensor is never stored when it is large"""
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fused_ffn_quantized(
        ctx._h, _t(input, keep), _t(w1_int8, keep), _t(w1_scale, keep), _t(w1_zero, keep), _t(b1, keep), _t(w2_int8, keep),
        _t(w2_scale, keep), _t(w2_zero, keep), _t(b2, keep), C.c_int(int(apply_relu2)), _t(res1, keep), _t(res2, keep), out._h,
        sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def fused_ffn_quantized_ln(input, w1_int8, w1_scale, w1_zero, b1, w2_int8, w2_scale, w2_zero, b2, apply_relu2, res1, res2, ln_scale, ln_bias,
                           epsilon, outs=None, ctx=None):
    """(y, layer_norm(y, ln_scale, ln_bias, -1, epsilon)) with y = fused_ffn_quantized(...): the feed-forward block, its residual Adds and
    the next half-layer's LayerNorm as one call (bit for bit the two calls)"""
    ctx = _ctx(ctx)
    keep = []
    outs = list(outs) if outs else [ctx.buf(), ctx.buf()]
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fused_ffn_quantized_ln(
        ctx._h, _t(input, keep), _t(w1_int8, keep), _t(w1_scale, keep), _t(w1_zero, keep), _t(b1, keep), _t(w2_int8, keep),
        _t(w2_scale, keep), _t(w2_zero, keep), _t(b2, keep), C.c_int(int(apply_relu2)), _t(res1, keep), _t(res2, keep), _t(ln_scale, keep),
        _t(ln_bias, keep), C.c_float(float(epsilon)), outs[0]._h, outs[1]._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(outs[0], sh.get(), np.float32)), TensorView(_lib.DevTensor(outs[1], sh.get(), np.float32))


def softmax_scaled(input, scale, axis=-1, out=None, ctx=None):
    """softmax(input * scale[0]): bit-identical to mul(input, scale) followed by softmax"""
    return _op(ctx, _lib.lib().lele_hip_softmax_scaled, [input, scale], [C.c_int32(axis)], out)


def add3(a, b, c, out=None, ctx=None):
    """(a + b) + c on equal shapes: bit-identical to add(add(a, b), c)"""
    return _op(ctx, _lib.lib().lele_hip_add3, [a, b, c], [], out)


def halves_pow_add_sqrt(input, axis, lo, hi, exp_lo, exp_hi, out=None, ctx=None):
    """sqrt(pow(x[.., lo[0]:lo[1], ..], exp_lo) + pow(x[.., hi[0]:hi[1], ..], exp_hi)) along axis (Slice's bound rules): bit-identical
    to sqrt(add(pow(slice(..)), pow(slice(..)))) -- the magnitude of a spectrum stored as [re | im] channel halves"""
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_halves_pow_add_sqrt(ctx._h, _lib.as_tensor(unwrap(input), keep), C.c_int32(axis), C.c_int64(lo[0]),
                                                       C.c_int64(lo[1]), C.c_int64(hi[0]), C.c_int64(hi[1]),
                                                       _lib.as_tensor(unwrap(exp_lo), keep), _lib.as_tensor(unwrap(exp_hi), keep),
                                                       out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def depthwise_conv1d_tlc(input, weights, bias=None, pad_left=0, pad_right=0, relu=False, x_offset=0, add_input=False, out=None, ctx=None):
    """transpose(0,2,1) -> depthwise conv1d -> transpose(0,2,1) on a time-major tensor, without the transposes; reads the
    channels [x_offset, x_offset + C) of input [B, T, P] in place; add_input adds the convolved input (the FSMN residual)"""
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_depthwise_conv1d_tlc(
        ctx._h, _lib.as_tensor(unwrap(input), keep), C.c_int64(int(x_offset)), _lib.as_tensor(unwrap(weights), keep),
        _lib.as_tensor(unwrap(bias), keep), C.c_int64(int(pad_left)), C.c_int64(int(pad_right)), C.c_int(int(bool(relu))),
        C.c_int(int(bool(add_input))), out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))

# ------------------------------------------------------------------------------------------- ConvInteger family
def conv_integer(input, weights, x_zero_point=None, w_zero_point=None, dilations=(), group=1, pads=(), strides=(), out=None,
                 ctx=None):
    """conv2d.rs:2216: u8 values carried as f32; zero points are [1] host tensors or None"""
    keep = []
    args = []
    d, n = _lib.i64_array(list(dilations), keep)
    args += [d, n, C.c_int64(int(group))]
    for v in (pads, strides):
        a, n = _lib.i64_array(list(v), keep)
        args += [a, n]
    return _op(ctx, _lib.lib().lele_hip_conv_integer, [input, weights, x_zero_point, w_zero_point], args, out)


def _conv_integer_from(sources, weights, w_zero_point, dilations, group, pads, strides, out, ctx):
    ctx = _ctx(ctx)
    keep = []
    ptrs = [_lib.as_tensor(unwrap(t), keep) for t in sources]
    arr = (C.POINTER(_lib.LeleTensor) * len(ptrs))(*[C.cast(p, C.POINTER(_lib.LeleTensor)) for p in ptrs])
    out = out or ctx.buf()
    osc = ctx.buf()
    sh = _lib.OutShape()
    d, nd = _lib.i64_array(list(dilations), keep)
    p, npd = _lib.i64_array(list(pads), keep)
    st, ns = _lib.i64_array(list(strides), keep)
    _lib.check(_lib.lib().lele_hip_conv_integer_from_f32(
        ctx._h, arr, C.c_size_t(len(ptrs)), _lib.as_tensor(unwrap(weights), keep), _lib.as_tensor(unwrap(w_zero_point), keep),
        d, nd, C.c_int64(int(group)), p, npd, st, ns, out._h, osc._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32)), TensorView(_lib.DevTensor(osc, (1,), np.float32))


def conv_integer_from_f32(input, weights, w_zero_point=None, dilations=(), group=1, pads=(), strides=(), out=None, ctx=None):
    """conv2d.rs:2246 -> (conv output, input scale [1] on the device)"""
    return _conv_integer_from([input], weights, w_zero_point, dilations, group, pads, strides, out, ctx)


def conv_integer_from_f32_multi(sources, weights, w_zero_point=None, out=None, ctx=None):
    """conv2d.rs:2420: channel-concatenated sources, joint dynamic range, 1x1 convolution"""
    return _conv_integer_from(list(sources), weights, w_zero_point, [1, 1], 1, [0, 0, 0, 0], [1, 1], out, ctx)


def fused_scale_bias(data, scale, bias, silu=False, scale_mul=1.0, out=None, ctx=None):
    """conv2d.rs:2710 / 2636 (silu=True): data * scale + bias[c]; `scale` is a python float or the [1] device tensor that
    conv_integer_from_f32 returned (then the effective scale is scale[0] * scale_mul)"""
    if isinstance(scale, (int, float, np.floating)):
        sdev, mul = None, float(scale) * float(scale_mul)
    else:
        sdev, mul = scale, float(scale_mul)
    return _fsb(ctx, data, sdev, mul, bias, silu, out)


def _fsb(ctx, data, sdev, mul, bias, silu, out):
    ctx = _ctx(ctx)
    keep = []
    out = out or ctx.buf()
    sh = _lib.OutShape()
    _lib.check(_lib.lib().lele_hip_fused_scale_bias(ctx._h, _lib.as_tensor(unwrap(data), keep), _lib.as_tensor(unwrap(sdev), keep),
                                                    C.c_float(mul), _lib.as_tensor(unwrap(bias), keep), C.c_int(int(bool(silu))),
                                                    out._h, sh.shape, C.byref(sh.rank)))
    return TensorView(_lib.DevTensor(out, sh.get(), np.float32))


def fused_scale_bias_silu(data, scale, bias, scale_mul=1.0, out=None, ctx=None):
    return fused_scale_bias(data, scale, bias, True, scale_mul, out, ctx)


# ------------------------------------------------------------------------------------------- i64-result comparisons
def _is_i64(x):
    return _dtype_of(x) == np.int64


def equal_i64(a, b, out=None, ctx=None):
    """math.rs:1201: element-wise a == b with numpy broadcasting, result 0/1 as i64 (inputs f32 or i64)"""
    if _is_i64(a) and _is_i64(b):
        return equal(a, b, out=out, ctx=ctx)
    return cast_to_i64(equal(a, b, ctx=ctx), out=out, ctx=ctx)


def equal_i64_f32_r(a, b, out=None, ctx=None):
    """math.rs:1209: both f32 operands are truncated to i64 (`v as i64`) before the comparison"""
    return equal(cast_to_i64(a, ctx=ctx), cast_to_i64(b, ctx=ctx), out=out, ctx=ctx)


def equal_i64_f32_r_i64(a, b, out=None, ctx=None):
    """math.rs:1219: a is i64, b is f32 truncated to i64"""
    return equal(a, cast_to_i64(b, ctx=ctx), out=out, ctx=ctx)


def equal_i64_f32_lhs(a, b, out=None, ctx=None):
    """math.rs:1228: a is f32 truncated to i64, b is i64"""
    return equal(cast_to_i64(a, ctx=ctx), b, out=out, ctx=ctx)


def less_i64(a, b, out=None, ctx=None):
    """math.rs:2161: a < b, result 0/1 as i64"""
    if _is_i64(a) and _is_i64(b):
        return less(a, b, out=out, ctx=ctx)
    return cast_to_i64(less(a, b, ctx=ctx), out=out, ctx=ctx)


def min_max(input, ctx=None):
    """math.rs:56: (min, max) of all elements as python floats (+inf / -inf for an empty tensor)"""
    shp = _shape_of(input)
    if int(np.prod(shp)) == 0:
        return float("inf"), float("-inf")
    axes = list(_b.range(len(shp)))
    mn = _reduce(4, input, axes, False, None, ctx).numpy().reshape(-1)[0]
    mx = _reduce(2, input, axes, False, None, ctx).numpy().reshape(-1)[0]
    return float(mn), float(mx)


add_f32 = add  # math.rs:366: the same-shape f32 fast path of add
