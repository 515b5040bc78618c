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
    if op in ("max", "min"):  # math.rs:1745-1756: `if val > *p { *p = val }` from -inf in input order -- NaN never wins, the first of equal values stays
        out = np.full(xt.shape[0], -np.inf if op == "max" else np.inf, np.float32)
        with np.errstate(invalid="ignore"):
            for j in range(n):
                col = xt[:, j]
                out = np.where(col > out if op == "max" else col < out, col, out)
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


# ---- data movement (bit-exact): manipulation.rs, shape.rs, conv2d.rs:1051-1502, math.rs:2033-2302 ----------------
def slice_(x, starts, ends, axes=(), steps=()):
    """manipulation.rs:258-380, restated step by step (NOT numpy slicing: an `end` below -dim clamps to -dim and then
    normalises to 0, so x[-2:-3:-1] on a length-2 axis is empty here, while numpy / ONNX yield one element)"""
    x = np.asarray(x)
    nd = x.ndim
    a_start, a_end, a_step = [0] * nd, list(x.shape), [1] * nd
    big = 2 ** 63 - 1
    for i in range(len(starts)):
        ax = i if not len(axes) else (axes[i] + nd if axes[i] < 0 else axes[i])
        dim = x.shape[ax]
        step = int(steps[i]) if i < len(steps) else 1
        s64, e64 = int(starts[i]), int(ends[i])
        e_max, e_min = e64 > big // 2, e64 < -(2 ** 63) // 2
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
            s, e = min(max(ns, 0), dim), min(max(ne, 0), dim)
        else:
            s, e = min(max(ns, 0), dim - 1), min(max(ne, -1), dim - 1)
        a_start[ax], a_end[ax], a_step[ax] = s, e, step
    idx = []
    for s, e, st in zip(a_start, a_end, a_step):
        if st > 0:
            cnt = 0 if s >= e else (e - s + st - 1) // st
        else:
            cnt = 0 if s <= e else (s - e + (-st) - 1) // (-st)
        idx.append(s + st * np.arange(cnt, dtype=np.int64))
    return x[np.ix_(*idx)] if nd else x


def pad(x, pads, value=0, mode="constant"):
    x = np.asarray(x)
    r = x.ndim
    raw = [max(0, int(p)) for p in pads]
    if len(raw) < 2 * r:
        half = len(raw) // 2
        full = [0] * (2 * r)
        for i in range(half):
            full[r - half + i] = raw[i]
            full[2 * r - half + i] = raw[half + i]
        raw = full
    if mode == "constant":
        return np.pad(x, [(raw[i], raw[r + i]) for i in range(r)], constant_values=value)
    if mode == "edge":
        return np.pad(x, [(raw[i], raw[r + i]) for i in range(r)], mode="edge")
    # reflect AS IMPLEMENTED by the reference (manipulation.rs:562-569): no edge repeat in front, edge repeated behind
    out = x
    for d in range(r):
        n = out.shape[d]
        src = [(-(c - raw[d])) if c < raw[d] else ((2 * n - 1 - (c - raw[d])) if c - raw[d] >= n else c - raw[d])
               for c in range(n + raw[d] + raw[r + d])]
        out = np.take(out, src, axis=d)
    return out


def gather(data, indices, axis):
    return np.take(np.asarray(data), np.asarray(indices).astype(np.int64), axis=axis)


def gather_elements(x, idx, axis):
    x = np.asarray(x)
    i = np.asarray(idx).astype(np.int64)
    i = np.where(i < 0, i + x.shape[axis], i)
    return np.take_along_axis(x, i, axis=axis)


def resize_nearest(x, out_h, out_w, asymmetric=True):  # conv2d.rs:1348-1377, f32 coordinate arithmetic
    x = np.asarray(x, np.float32)
    in_h, in_w = x.shape[2:]
    hs, ws = np.float32(in_h) / np.float32(out_h), np.float32(in_w) / np.float32(out_w)

    def rnd(v):  # f32::round, half away from zero
        return np.sign(v) * np.floor(np.abs(v) + np.float32(0.5))

    def idx(n_out, scale, n_in):
        o = np.arange(n_out, dtype=np.float32)
        if asymmetric:
            v = np.minimum(np.floor((o * scale).astype(np.float32)), np.float32(n_in - 1))
        else:
            t = ((o + np.float32(0.5)).astype(np.float32) * scale).astype(np.float32) - np.float32(0.5)
            v = np.minimum(np.maximum(rnd(t.astype(np.float32)), 0), np.float32(n_in - 1))
        return v.astype(np.int64)

    return x[:, :, idx(out_h, hs, in_h)][:, :, :, idx(out_w, ws, in_w)]


def max_pool2d(x, kernel, strides=(), pads=(), dilations=(), ceil_mode=False):  # conv2d.rs:1051-1254
    x = np.asarray(x, np.float32)
    n, c, ih, iw = x.shape
    kh = kernel[0]
    kw = kernel[1] if len(kernel) > 1 else kh
    sh = strides[0] if len(strides) else 1
    sw = strides[1] if len(strides) > 1 else sh
    pt = pads[0] if len(pads) else 0
    pl = pads[1] if len(pads) > 1 else pt
    pb = pads[2] if len(pads) > 2 else pt
    pr = pads[3] if len(pads) > 3 else pl
    dh = dilations[0] if len(dilations) else 1
    dw = dilations[1] if len(dilations) > 1 else dh
    ekh, ekw = dh * (kh - 1) + 1, dw * (kw - 1) + 1
    nh, nw = ih + pt + pb - ekh, iw + pl + pr - ekw
    oh = (nh + sh - 1) // sh + 1 if ceil_mode else nh // sh + 1
    ow = (nw + sw - 1) // sw + 1 if ceil_mode else nw // sw + 1
    xp = np.full((n, c, ih + pt + pb + sh * 2 + ekh, iw + pl + pr + sw * 2 + ekw), -np.inf, np.float32)
    xp[:, :, pt:pt + ih, pl:pl + iw] = x
    out = np.full((n, c, oh, ow), -np.inf, np.float32)
    for a in range(kh):
        for b in range(kw):
            out = np.maximum(out, xp[:, :, a * dh:a * dh + oh * sh:sh, b * dw:b * dw + ow * sw:sw])
    return out


def topk(x, k, largest=True):  # conv2d.rs:1385-1435: stable sort, indices as f32
    x = np.asarray(x, np.float32)
    k = min(k, x.shape[-1])
    order = np.argsort(-x if largest else x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(x, order, -1), order.astype(np.float32)


def wav_to_f32(payload, bits_per_sample=16, num_channels=1):
    """examples/sensevoice/src/audio.rs:52-73"""
    b = np.frombuffer(bytes(payload), np.uint8)
    if bits_per_sample == 16:
        s = b[:len(b) // 2 * 2].view("<i2").astype(np.float32) / np.float32(32768.0)
    elif bits_per_sample == 8:
        s = (b.astype(np.float32) - np.float32(128.0)) / np.float32(128.0)
    else:
        raise ValueError("Unsupported bits per sample: %d" % bits_per_sample)
    if num_channels == 2:
        s = (s[0::2] + s[1::2]) / np.float32(2.0)
    return s.astype(np.float32)


def argmax_last(x):
    """tokenizer.rs:50-61: Iterator::max_by keeps the last of equal maxima"""
    x = np.asarray(x, np.float32)
    v = x.shape[-1]
    return (v - 1 - np.argmax(x[..., ::-1], axis=-1)).astype(np.int32)


# ---- ConvInteger family (conv2d.rs:1507-2761, x86 branches); the convolution itself is pyoracle.conv2d
def dql_params(xs):
    """scale, zp of the joint range of the arrays (conv2d.rs:2329-2333), in f32 arithmetic"""
    mn = np.float32(min(float(np.min(x)) for x in xs if np.size(x)))
    mx = np.float32(max(float(np.max(x)) for x in xs if np.size(x)))
    amin, amax = np.float32(min(mn, np.float32(0))), np.float32(max(mx, np.float32(0)))
    rng = np.float32(max(np.float32(amax - amin), np.float32(1e-5)))
    scale = np.float32(rng / np.float32(255.0))
    z = np.float32(-amin) / scale
    zp = np.float32(np.clip(np.sign(z) * np.floor(np.abs(z) + np.float32(0.5)), 0, 255))  # f32::round: half away from zero
    return scale, zp


def dql_quantize(x, scale, zp):
    inv = np.float32(1.0) / scale
    t = (np.asarray(x, np.float32) * inv).astype(np.float32) + zp
    r = np.sign(t) * np.floor(np.abs(t) + np.float32(0.5))
    return np.clip(r, 0, 255).astype(np.float32)


def fused_scale_bias(data, scale, bias, silu=False):
    x = (np.asarray(data, np.float32) * np.float32(scale)).astype(np.float32) + np.asarray(bias, np.float32).reshape(1, -1, 1, 1)
    x = x.astype(np.float32)
    return (x / (np.float32(1.0) + np.exp(-x).astype(np.float32))).astype(np.float32) if silu else x


def adaptive_avg_pool1d(x, output_len):  # pooling.rs:1-30 (sequential f32 sum over the window, then / len)
    x = np.asarray(x, np.float32)
    L = x.shape[-1]
    flat = x.reshape(-1, L)
    out = np.zeros((flat.shape[0], output_len), np.float32)
    for i in range(output_len):
        start = min((i * L) // output_len, L)
        end = min(-(-((i + 1) * L) // output_len), L)
        if end - start == 0:
            continue
        acc = np.zeros(flat.shape[0], np.float32)
        for k in range(start, end):
            acc = (acc + flat[:, k]).astype(np.float32)
        out[:, i] = acc / np.float32(end - start)
    return out.reshape(x.shape[:-1] + (output_len,))


def depthwise_conv2d_x86_generic(x, w, strides, pads, relu=False):
    """TEST INFRASTRUCTURE.  What lele's x86 build computes for a depthwise convolution that is NOT the 3x3 / stride 1 / pad 1
    special case: /root/reference/src/kernels/conv2d.rs:535-570 dispatches to depthwise_conv2d_avx2 (3130-3215) WITHOUT the
    bias and with only ReLU honoured (SiLU is dropped), and that kernel's 8-wide middle reads `ow + kj - pad_left + lane` without
    a right-edge test (3156-3163: `in_w + pad_left.saturating_sub(kw - 1 + 7)` is in_w for every usual padding), so the last
    vector step of a row takes pixels of the NEXT row (flat memory) where the definition has zero padding.  Restated here so that
    the device library's deliberate divergence (it implements the ONNX definition: DESIGN.md section 4) is pinned by a test
    instead of being silent.  Returns (out [N,C,OH,OW], defined): `defined` is False where the x86 code reads past the end of
    the input buffer (undefined behaviour upstream).  FMA is emulated in f64 (exact product, one extra rounding): compare at 1e-6."""
    x = np.ascontiguousarray(x, np.float32)
    w = np.ascontiguousarray(w, np.float32)
    n, c, ih, iw = x.shape
    kh, kw = w.shape[-2:]
    sh, sw = strides
    pt, pl, pb, pr = pads
    oh, ow = (ih + pt + pb - kh) // sh + 1, (iw + pl + pr - kw) // sw + 1
    flat = x.reshape(-1)
    out = np.zeros((n, c, oh, ow), np.float32)
    defined = np.ones((n, c, oh, ow), bool)
    vec_end = ((iw + max(pl - (kw - 1 + 7), 0)) // 8) * 8 if sw == 1 else 0
    vec_start = pl if sw == 1 else ow

    def scalar_px(b, ch, y, xo):
        s = np.float32(0)
        for ki in range(kh):
            yy = y * sh + ki - pt
            if yy < 0 or yy >= ih:
                continue
            for kj in range(kw):
                xx = xo * sw + kj - pl
                if xx < 0 or xx >= iw:
                    continue
                s = np.float32(s + x[b, ch, yy, xx] * w[ch].reshape(kh, kw)[ki, kj])
        return np.float32(0) if (relu and s < 0) else s
    for b in range(n):
        for ch in range(c):
            base = (b * c + ch) * ih * iw
            wk = w[ch].reshape(kh, kw)
            for y in range(oh):
                xo = 0
                while xo < min(vec_start, ow):
                    out[b, ch, y, xo] = scalar_px(b, ch, y, xo)
                    xo += 1
                while sw == 1 and xo + 8 <= vec_end and xo + 8 <= ow:
                    acc = np.zeros(8, np.float64)
                    ok = np.ones(8, bool)
                    for ki in range(kh):
                        yy = y * sh + ki
                        if yy < pt or yy >= ih + pt:
                            continue
                        row = base + (yy - pt) * iw
                        for kj in range(kw):
                            idx = row + xo + kj - pl + np.arange(8)
                            inb = idx < flat.size
                            ok &= inb
                            v = flat[np.minimum(idx, flat.size - 1)].astype(np.float64)
                            acc = (np.float64(wk[ki, kj]) * v + acc).astype(np.float32).astype(np.float64)
                    r = acc.astype(np.float32)
                    if relu:
                        r = np.maximum(r, np.float32(0))
                    out[b, ch, y, xo:xo + 8] = r
                    defined[b, ch, y, xo:xo + 8] = ok
                    xo += 8
                while xo < ow:
                    out[b, ch, y, xo] = scalar_px(b, ch, y, xo)
                    xo += 1
    return out, defined
