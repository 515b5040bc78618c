"""numpy restatement of lele's index / single-IEEE-operation kernels -- TEST INFRASTRUCTURE (see oracle/oracle.h).

These ops perform at most ONE rounding per output element (or pure data movement), so numpy's float32 arithmetic
IS the reference arithmetic; what is restated here is the reference's shape/broadcast/attribute semantics.
Citations are file:line under /root/reference.
"""
import numpy as np


# ---- src/kernels/math.rs: broadcast_binary_op (69-264) and the ops built on it --------------------------------
def binary(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    with np.errstate(all="ignore"):
        if name == "add":
            return a + b  # math.rs:414
        if name == "sub":
            return a - b  # math.rs:838
        if name == "mul":
            return a * b  # math.rs:611
        if name == "div":
            if a.dtype == np.int64:
                return np.where(b == 0, 0, np.trunc(a / np.where(b == 0, 1, b))).astype(np.int64)
            return a / b  # math.rs:1106
        if name == "pow":
            return np.power(a.astype(np.float64), b.astype(np.float64)).astype(np.float32)  # f32::powf (libm)
        if name == "max":
            return np.fmax(a, b)  # f32::max: NaN-ignoring, math.rs:1922
        if name == "min":
            return np.fmin(a, b)
        if name == "equal":
            return (a == b).astype(a.dtype)  # 1.0 / 0.0, math.rs:1193
        if name == "less":
            return (a < b).astype(a.dtype)  # math.rs:2154
        if name == "greater":
            return (a > b).astype(a.dtype)
        if name == "prelu":
            return np.where(a < 0, a * b, a).astype(np.float32)  # math.rs:2012-2031
        if name == "mod_f32":
            q = np.floor(a / np.where(b == 0, 1, b)).astype(np.float32)
            return np.where(b == 0, np.float32(0), a - (b * q).astype(np.float32)).astype(np.float32)  # math.rs:1163
        if name == "and_":
            return ((a != 0) & (b != 0)).astype(np.float32)
        if name == "or_":
            return ((a != 0) | (b != 0)).astype(np.float32)
    raise KeyError(name)


def where_op(cond, x, y):  # manipulation.rs:1215-
    return np.where(np.asarray(cond) != 0, x, y).astype(np.float32)


def clip(x, lo=None, hi=None):  # math.rs:1984-2010
    x = np.asarray(x, np.float32)
    lo = np.float32(-3.40282347e+38) if lo is None else np.float32(lo)
    hi = np.float32(3.40282347e+38) if hi is None else np.float32(hi)
    return np.minimum(np.maximum(x, lo), hi)


def unary_exact(name, x):
    """the unary kernels without a SIMD body (one correctly rounded operation, or libm): math.rs:893-905, 1046-1056,
    1508-1525, 2084-2153"""
    x = np.asarray(x, np.float32)
    with np.errstate(all="ignore"):
        if name == "neg":
            return -x
        if name == "reciprocal":
            return (np.float32(1.0) / x).astype(np.float32)
        if name == "not_":
            return (x == 0).astype(np.float32)
        if name == "abs":
            return np.abs(x)
        if name == "floor":
            return np.floor(x)
        if name == "ceil":
            return np.ceil(x)
        x64 = x.astype(np.float64)
        if name == "log":
            return np.log(x64).astype(np.float32)
        if name == "sin":
            return np.sin(x64).astype(np.float32)
        if name == "cos":
            return np.cos(x64).astype(np.float32)
        if name == "softplus":
            return np.where(x > 20.0, x, np.log1p(np.exp(x64))).astype(np.float32)
    raise KeyError(name)


# ---- reductions, math.rs:1527-1920: accumulate in row-major INPUT order, one rounding per add --------------------
def reduce(op, x, axes, keepdims):
    x = np.asarray(x, np.float32)
    dims = x.ndim
    ax = sorted({a + dims if a < 0 else a for a in axes}) or list(range(dims))
    keep = [d for d in range(dims) if d not in ax]
    perm = keep + ax
    xt = np.transpose(x, perm).reshape(int(np.prod([x.shape[d] for d in keep], dtype=np.int64)), -1)
    n = xt.shape[1]
    if op == "max":
        out = xt.max(axis=1) if n else np.full(xt.shape[0], -np.inf, np.float32)
    elif op == "min":
        out = xt.min(axis=1)
    else:
        acc = np.zeros(xt.shape[0], np.float32)
        for j in range(n):  # sequential f32 accumulation in input order
            acc = (acc + (xt[:, j] * xt[:, j] if op == "l2" else xt[:, j])).astype(np.float32)
        if op == "mean":
            acc = (acc * np.float32(np.float32(1.0) / np.float32(n))).astype(np.float32)
        if op == "l2":
            acc = np.sqrt(acc)
        out = acc
    oshape = [1 if d in ax else x.shape[d] for d in range(dims)] if keepdims else [x.shape[d] for d in keep]
    return out.astype(np.float32).reshape(oshape)
