"""Shape arithmetic on the host.

ONNX graphs carry small int64 side computations (Shape -> Gather -> Concat -> Reshape target ...).  lele folds them at
compile time where the operands are constant (src/compiler/mod.rs:375-760) and otherwise runs them through its kernels on
`Cow<[i64]>` data.  Here they never touch the device: the constant folder (lower.py) and the plan runner (plan.py) both
evaluate them with the functions below whenever every operand is a host integer array, so that a captured hipGraph holds
only the real tensor work.  Integer semantics only -- nothing here is numerically delicate.
"""
import numpy as np


def _axes(v):
    return [int(a) for a in np.asarray(v).reshape(-1)]


def unsqueeze(x, axes):
    x = np.asarray(x)
    out_rank = x.ndim + len(axes)
    for a in sorted(a + out_rank if a < 0 else a for a in _axes(axes)):
        x = np.expand_dims(x, a)
    return x


def squeeze(x, axes=None):
    x = np.asarray(x)
    if axes is None or len(_axes(axes)) == 0:
        return x.reshape([d for d in x.shape if d != 1])
    return np.squeeze(x, tuple(a + x.ndim if a < 0 else a for a in _axes(axes)))


def slice_(x, starts, ends, axes=None, steps=None):
    """ONNX Slice on a host array (negative indices wrap once, then clamp; INT64 sentinels clamp)"""
    x = np.asarray(x)
    starts, ends = _axes(starts), _axes(ends)
    axes = _axes(axes) if axes is not None and len(_axes(axes)) else list(range(len(starts)))
    steps = _axes(steps) if steps is not None and len(_axes(steps)) else [1] * len(starts)
    idx = [slice(None)] * x.ndim
    for s, e, a, st in zip(starts, ends, axes, steps):
        a = a + x.ndim if a < 0 else a
        d = x.shape[a]
        s = s + d if s < 0 else s
        e = e + d if e < 0 else e
        if st > 0:
            idx[a] = slice(min(max(s, 0), d), min(max(e, 0), d), st)
        else:
            s, e = min(max(s, 0), d - 1), min(max(e, -1), d - 1)
            idx[a] = slice(s, None if e < 0 else e, st)
    return x[tuple(idx)]


def gather(x, indices, axis=0):
    x, ind = np.asarray(x), np.asarray(indices).astype(np.int64)
    axis = axis + x.ndim if axis < 0 else axis
    ind = np.where(ind < 0, ind + x.shape[axis], ind)
    return np.take(x, ind, axis=axis)


def cast(x, to):
    from . import onnx_pb as pb
    x = np.asarray(x)
    if to in (pb.INT64, pb.INT32, pb.BOOL):  # lele carries every integer / boolean tensor as i64 (ops/tensor.rs Cast)
        return x.astype(np.int64)
    if to in (pb.FLOAT, pb.FLOAT16, pb.DOUBLE):
        return x.astype(np.float32)
    return x


def range_(start, limit, delta):
    s, l, d = (np.asarray(v).reshape(-1)[0] for v in (start, limit, delta))
    return np.arange(s, l, d).astype(np.asarray(start).dtype)


def constant_of_shape(shape, value):
    return np.full(_axes(shape), value)


def expand(x, shape):
    x = np.asarray(x)
    tgt = _axes(shape)
    return x * np.ones(tgt, x.dtype) if len(tgt) else x


_BINARY = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Max": np.maximum, "Min": np.minimum}


def binary(op, a, b):
    a, b = np.asarray(a), np.asarray(b)
    if op == "Div":
        if a.dtype.kind in "iu" and b.dtype.kind in "iu":  # truncating integer division (Rust `/`), divisor 0 -> 0
            q = np.trunc(a.astype(np.float64) / np.where(b == 0, 1, b).astype(np.float64)).astype(np.int64)
            return np.where(b == 0, 0, q)
        return a / b
    if op in ("Equal", "Less", "Greater"):
        r = {"Equal": np.equal, "Less": np.less, "Greater": np.greater}[op](a, b)
        return r.astype(np.int64)
    return _BINARY[op](a, b)


def evaluate(op, inputs, attrs):
    """one ONNX node on host arrays -> list of outputs, or None when the op is not a host shape op"""
    x = inputs
    if op == "Identity":
        return [np.asarray(x[0])]
    if op == "Shape":
        return [np.array(np.asarray(x[0]).shape, np.int64)]
    if op == "Size":
        return [np.array(np.asarray(x[0]).size, np.int64)]
    if op == "Unsqueeze":
        return [unsqueeze(x[0], x[1] if len(x) > 1 and x[1] is not None else attrs.get("axes", []))]
    if op == "Squeeze":
        return [squeeze(x[0], x[1] if len(x) > 1 and x[1] is not None else attrs.get("axes"))]
    if op == "Concat":
        return [np.concatenate([np.atleast_1d(v) for v in x], axis=attrs.get("axis", 0))]
    if op == "Gather":
        return [gather(x[0], x[1], attrs.get("axis", 0))]
    if op == "Cast":
        return [cast(x[0], attrs["to"])]
    if op in ("Add", "Sub", "Mul", "Div", "Equal", "Less", "Greater", "Max", "Min"):
        return [binary(op, x[0], x[1])]
    if op == "Neg":
        return [-np.asarray(x[0])]
    if op == "Reshape":
        a = np.asarray(x[0])
        tgt = [a.shape[i] if d == 0 and i < a.ndim else d for i, d in enumerate(_axes(x[1]))]
        return [a.reshape(tgt)]
    if op == "Slice":
        g = lambda i: x[i] if len(x) > i and x[i] is not None else None  # noqa: E731
        return [slice_(x[0], x[1], x[2], g(3), g(4))]
    if op == "Range":
        return [range_(x[0], x[1], x[2])]
    if op == "ConstantOfShape":
        return [constant_of_shape(x[0], attrs.get("value", np.float32(0.0)))]
    if op == "Expand":
        return [expand(x[0], x[1])]
    if op == "Where":
        return [np.where(np.asarray(x[0]) != 0, x[1], x[2])]
    if op == "Transpose":
        return [np.transpose(np.asarray(x[0]), attrs.get("perm") or None)]
    if op == "Tile":
        return [np.tile(np.asarray(x[0]), _axes(x[1]))]
    if op == "Not":
        return [(np.asarray(x[0]) == 0).astype(np.int64)]
    return None
