"""Concat / Split along C as channel views (lele_amd.plan.fold_channel_views, the *_pitched entry points of the C ABI).

lele copies (manipulation.rs:108-207 concat, 1091-1151 split); a folded plan moves no value, so it must give the bits of the plan
it came from.  CPU half: the pass, the slot allocator and the runner's view handling on plans of the shapes a Yolo-style network
produces (C3k2: conv -> split -> bottleneck -> concat -> conv; SPPF: conv -> three chained max-pools -> concat; FPN: resize +
concat), executed by an EMULATION of the kernels on buffer-faithful memory (one flat array per workspace slot, re-allocated --
and poisoned -- when it grows; values are windows of it) with the oracle's arithmetic: a value clobbered before its last reader,
a window landing in the wrong place or a view handed to an operator that cannot take one all show up as wrong numbers.
GPU half: the same plans through the library, folded == unfolded bit for bit, and the pitched kernels against the oracle."""
import numpy as np
import pytest

from lele_amd import _lib
from lele_amd.plan import Runner, fold_channel_views
from lele_amd.tensor import TensorView
from oracle import npref
from oracle import pyoracle as O


# --------------------------------------------------------------------------------------------- buffer-faithful emulation (CPU)
class EmuBuf:
    def __init__(self):
        self.mem = np.zeros(0, np.float32)

    def reserve(self, nbytes):
        n = (int(nbytes) + 3) // 4
        if n > self.mem.size:            # LeleBuf::reserve re-allocates without keeping the contents
            self.mem = np.full(n, np.nan, np.float32)

    def to_numpy(self, shape, dtype=np.float32):
        n = int(np.prod(shape, dtype=np.int64))
        assert n <= self.mem.size, "read past the buffer"
        return self.mem[:n].copy().reshape(shape).astype(dtype)

    def close(self):
        pass


class EmuCtx:
    def buf(self):
        return EmuBuf()

    def sync(self):
        pass


def _val(x):
    if isinstance(x, TensorView):
        return x.numpy()
    if isinstance(x, _lib.Weight):
        return x.arr
    return None if x is None else np.asarray(x)


class EmuK:
    """the operators of the test plans with the oracle's arithmetic, writing where the library would write"""
    VIEW_OK = {"conv2d", "conv2d_silu", "conv2d_fused", "conv2d_res", "add", "mul", "max_pool2d", "resize_nearest", "copy_view", "transpose_cp"}

    def __init__(self):
        self.view_reads = 0

    def _in(self, fn, *ts):
        for t in ts:
            if isinstance(t, TensorView) and t.is_view:
                assert fn in self.VIEW_OK, "%s was handed a channel view" % fn
                self.view_reads += 1

    @staticmethod
    def _put(res, out, out_window):
        res = np.ascontiguousarray(res, np.float32)
        n, per = res.shape[0], int(np.prod(res.shape[1:], dtype=np.int64))
        if out_window:
            off, pitch = out_window
            assert pitch >= per and off + (n - 1) * pitch + per <= out.mem.size, "window outside the reserved tensor"
            for i in range(n):
                out.mem[off + i * pitch:off + i * pitch + per] = res[i].reshape(-1)
            return TensorView(_lib.DevTensor(out, res.shape, np.float32, off, pitch))
        out.reserve(4 * res.size)
        out.mem[:res.size] = res.reshape(-1)
        return TensorView(_lib.DevTensor(out, res.shape, np.float32))

    def _conv(self, fn, act, x, w, b, dil, group, pads, strides, out, out_window):
        self._in(fn, x)
        return self._put(O.conv2d_im2col(_val(x), _val(w), _val(b), dil, group, pads, strides, act), out, out_window)

    def conv2d(self, x, w, b, dil, group, pads, strides, out=None, ctx=None, out_window=None):
        return self._conv("conv2d", None, x, w, b, dil, group, pads, strides, out, out_window)

    def conv2d_silu(self, x, w, b, dil, group, pads, strides, out=None, ctx=None, out_window=None):
        return self._conv("conv2d_silu", "silu", x, w, b, dil, group, pads, strides, out, out_window)

    def conv2d_res(self, x, w, b, res, dil, group, pads, strides, act, out=None, ctx=None, out_window=None):
        self._in("conv2d_res", x, res)
        y = O.conv2d_im2col(_val(x), _val(w), _val(b), dil, group, pads, strides, {0: None, 1: "relu", 2: "silu"}[act])
        self.res_calls = getattr(self, "res_calls", 0) + 1
        return self._put(y + _val(res), out, out_window)

    def add(self, a, b, out=None, ctx=None, out_window=None):
        self._in("add", a, b)
        return self._put(_val(a) + _val(b), out, out_window)

    def sigmoid(self, a, out=None, ctx=None):
        self._in("sigmoid", a)
        return self._put(O.unary("sigmoid", _val(a)), out, None)

    def max_pool2d(self, x, k, s, p, d, ceil, out=None, ctx=None, out_window=None):
        self._in("max_pool2d", x)
        return self._put(npref.max_pool2d(_val(x), k, s, p, d, ceil), out, out_window)

    def resize_nearest(self, x, scales=None, sizes=None, mode="asymmetric", out=None, ctx=None, out_window=None):
        self._in("resize_nearest", x)
        v = _val(x)
        return self._put(npref.resize_nearest(v, int(v.shape[2] * scales[2]), int(v.shape[3] * scales[3])), out, out_window)

    def copy_view(self, x, out=None, out_window=None, ctx=None):
        self._in("copy_view", x)
        return self._put(_val(x), out, out_window)

    def transpose_cp(self, x, out=None, out_window=None, ctx=None):
        self._in("transpose_cp", x)
        v = _val(x)
        self.tcp_calls = getattr(self, "tcp_calls", 0) + 1
        return self._put(np.ascontiguousarray(v.reshape(v.shape[0], v.shape[1], -1).transpose(0, 2, 1)), out, out_window)

    def reshape(self, x, shape):
        from lele_amd import kernels as K
        return K.reshape(x, shape)

    def transpose(self, x, perm, out=None, ctx=None):
        self._in("transpose", x)
        return self._put(np.ascontiguousarray(_val(x).transpose(perm)), out, None)

    def concat(self, xs, axis, out=None, ctx=None):
        self._in("concat", *xs)
        return self._put(np.concatenate([_val(x) for x in xs], axis), out, None)

    def split(self, x, axis, sizes, outputs=None, ctx=None):
        self._in("split", x)
        v, res, c0 = _val(x), [], 0
        for sz, ob in zip(sizes, outputs):
            res.append(self._put(np.take(v, range(c0, c0 + sz), axis), ob, None))
            c0 += sz
        return res


# --------------------------------------------------------------------------------------------- the test plans
class PlanBuilder:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.sts, self.weights, self.off, self.k = [], {}, 0, 0

    def w(self, arr):
        arr = np.asarray(arr, np.float32)
        node = ["weight_f32", self.off, arr.nbytes, list(arr.shape)]
        self.weights["%d:weight_f32:%s" % (self.off, "x".join(map(str, arr.shape)))] = (node, arr)
        self.off += arr.nbytes
        return {"weight": node}

    def name(self, tag):
        self.k += 1
        return "%s_%d" % (tag, self.k)

    def call(self, fn, args, tag, nout=1):
        outs = [self.name(tag) for _ in range(nout)]
        self.sts.append({"op": "call", "out": outs, "fn": fn, "args": args, "bufs": nout})
        return outs[0] if nout == 1 else outs

    @staticmethod
    def ints(v):
        return {"list": [{"int": int(x)} for x in v]}

    def conv(self, x, cin, cout, k=1, s=1, silu=True, group=1):
        w = self.rng.standard_normal((cout, cin // group, k, k)) * np.sqrt(2.0 / (cin // group * k * k))
        b = self.rng.standard_normal(cout) * 0.1
        return self.call("conv2d_silu" if silu else "conv2d",
                         [{"ref": x}, self.w(w), self.w(b), self.ints([1, 1]), {"int": group}, self.ints([k // 2] * 4), self.ints([s, s])], "conv")

    def finish(self, inputs, outputs):
        from lele_amd.compiler.lower import allocate
        slots = allocate(self.sts, outputs)
        plan = {"source": "test", "format": "lele_amd.plan/2", "inputs": inputs, "outputs": outputs, "slots": slots, "statements": self.sts,
                "weights": {k: v[0] for k, v in self.weights.items()}}
        return plan, {k: v[1] for k, v in self.weights.items()}


def c3k2_sppf_fpn_plan(seed=0):
    """x [N,16,H,W] -> C3k2 (conv, split, bottleneck with a residual add, concat of [y0, y1, m], conv) -> SPPF (conv, three
    chained 5x5 pools, concat, conv) -> upsample + concat with the C3k2 result + a grouped conv (must NOT take a view) + an output
    that is a split result (must stay a copy)"""
    b = PlanBuilder(seed)
    t = b.conv("x", 16, 32, 1)
    y0, y1 = b.call("split", [{"ref": t}, {"int": 1}, b.ints([16, 16])], "split", 2)
    m = b.conv(b.conv(y1, 16, 8, 3), 8, 16, 3)
    m = b.call("add", [{"ref": y1}, {"ref": m}], "add")
    cat = b.call("concat", [{"list": [{"ref": y0}, {"ref": y1}, {"ref": m}]}, {"int": 1}], "cat")
    p3 = b.conv(cat, 48, 32, 1)
    d = b.conv(p3, 32, 32, 3, 2)
    y = b.conv(d, 32, 16, 1)
    ps = [y]
    for _ in range(3):
        ps.append(b.call("max_pool2d", [{"ref": ps[-1]}, b.ints([5, 5]), b.ints([1, 1]), b.ints([2, 2, 2, 2]), b.ints([1, 1]), {"bool": False}], "pool"))
    sp = b.conv(b.call("concat", [{"list": [{"ref": p} for p in ps]}, {"int": 1}], "cat"), 64, 32, 1)
    up = b.call("resize_nearest", [{"ref": sp}, {"list": [{"float": 1.0}, {"float": 1.0}, {"float": 2.0}, {"float": 2.0}]}, {"none": 1},
                                   {"str": "asymmetric"}], "up")
    n3 = b.conv(b.call("concat", [{"list": [{"ref": up}, {"ref": p3}]}, {"int": 1}], "cat"), 64, 32, 3)
    # a second split whose results feed a GROUPED convolution (no view) and the graph output (no view): it stays a copy kernel
    q0, q1 = b.call("split", [{"ref": n3}, {"int": 1}, b.ints([16, 16])], "split", 2)
    g = b.conv(q0, 16, 16, 3, group=16)
    # a concat along H (not C): untouched
    tall = b.call("concat", [{"list": [{"ref": g}, {"ref": q1}]}, {"int": 2}], "cat")
    return b.finish(["x"], [tall, q1, sp])


def run_plan(plan, weights, ctx, K, feed, record=False):
    r = Runner(plan, weights, ctx)
    if K is not None:
        r.K = K
    if record:
        r.shapes = {}
    outs = [o.numpy().copy() for o in r.run(feed)]
    return outs, r


def test_fold_on_emulated_memory():
    plan, weights = c3k2_sppf_fpn_plan()
    x = np.random.default_rng(1).standard_normal((3, 16, 16, 24)).astype(np.float32)
    ctx = EmuCtx()

    def feed():
        return {"x": EmuK._put(x, ctx.buf(), None)}
    want, r0 = run_plan(plan, weights, ctx, EmuK(), feed(), record=True)
    folded = fold_channel_views(plan, r0.shapes)
    info = folded["folded"]
    # C3k2: the split's input goes into the concat in place (one merged operand) + the residual add; SPPF: the 1x1 conv and the three
    # pools; FPN: the resize and p3.  The split feeding the grouped conv / the output stays, as does the concat along H.
    # The bottleneck's `y1 + cv2(cv1(y1))` is ONE conv2d_res call that reads the split result as a view and writes the concat's window.
    assert info == {"residual_adds_fused": 1, "transposed_splits_folded": 0, "concats_in_place": 3, "splits_as_views": 1, "operands_in_place": 8,
                    "operands_copied": 0}, info
    fns = [st.get("fn") for st in folded["statements"] if st["op"] == "call"]
    assert fns.count("concat") == 1 and fns.count("split") == 1 and fns.count("add") == 0 and fns.count("conv2d_res") == 1
    assert fold_channel_views(plan, r0.shapes, residuals=False)["folded"]["residual_adds_fused"] == 0
    k = EmuK()
    got, _ = run_plan(folded, weights, ctx, k, feed())
    assert k.view_reads > 0 and k.res_calls == 1
    for a, b in zip(want, got):
        assert a.shape == b.shape and np.array_equal(a, b)
    # twice through the same runner (buffers already sized, as in a replay) and with another batch size's shapes recorded anew
    r = Runner(folded, weights, ctx)
    r.K = EmuK()
    for _ in range(2):
        got2 = [o.numpy().copy() for o in r.run(feed())]
        assert all(np.array_equal(a, b) for a, b in zip(want, got2))


def test_fold_keeps_what_it_cannot_prove():
    plan, weights = c3k2_sppf_fpn_plan()
    # without shapes nothing is folded; an operand that two concats want is produced in place once and copied once
    assert fold_channel_views(plan, {})["folded"]["concats_in_place"] == 0
    b = PlanBuilder(3)
    a = b.conv("x", 8, 8, 1)
    c = b.conv("x", 8, 8, 1)
    c1 = b.call("concat", [{"list": [{"ref": a}, {"ref": c}]}, {"int": 1}], "cat")
    c2 = b.call("concat", [{"list": [{"ref": c}, {"ref": a}]}, {"int": 1}], "cat")
    s = b.call("sigmoid", [{"ref": c}], "sig")      # a reader that cannot take a view: `c` must stay dense
    plan2, w2 = b.finish(["x"], [c1, c2, s])
    ctx = EmuCtx()
    x = np.random.default_rng(2).standard_normal((2, 8, 6, 8)).astype(np.float32)
    want, r0 = run_plan(plan2, w2, ctx, EmuK(), {"x": EmuK._put(x, ctx.buf(), None)}, record=True)
    folded = fold_channel_views(plan2, r0.shapes)
    # cat(a, c): a in place, c copied; cat(c, a): nothing in place, but a is a view by now -> two copy_view calls
    assert folded["folded"]["operands_in_place"] == 1 and folded["folded"]["operands_copied"] == 3, folded["folded"]
    got, _ = run_plan(folded, w2, ctx, EmuK(), {"x": EmuK._put(x, ctx.buf(), None)})
    assert all(np.array_equal(p, q) for p, q in zip(want, got))


def test_residual_adds_are_fused_only_where_that_is_the_same_program():
    def build(case):
        b = PlanBuilder(11)
        t = b.conv("x", 8, 8, 1)
        if case == "two readers":            # the convolution's result is read by the Add and by another statement
            c = b.conv(t, 8, 8, 3)
            y = b.call("add", [{"ref": t}, {"ref": c}], "add")
            z = b.call("sigmoid", [{"ref": c}], "sig")
            return b.finish(["x"], [y, z])
        if case == "residual later":         # the other operand does not exist yet where the convolution runs
            c = b.conv(t, 8, 8, 3)
            u = b.call("sigmoid", [{"ref": t}], "sig")
            return b.finish(["x"], [b.call("add", [{"ref": c}, {"ref": u}], "add")])
        if case == "broadcast":              # [N, 8, H, W] + [N, 8, 1, 1]
            c = b.conv(t, 8, 8, 3)
            pooled = b.call("max_pool2d", [{"ref": t}, b.ints([6, 8]), b.ints([6, 8]), b.ints([0, 0, 0, 0]), b.ints([1, 1]), {"bool": False}], "pool")
            return b.finish(["x"], [b.call("add", [{"ref": c}, {"ref": pooled}], "add")])
        if case == "both orders":            # add(r, conv) and add(conv, r); linear and SiLU; the second feeds the first's residual
            c1 = b.conv(t, 8, 8, 3, silu=False)
            y1 = b.call("add", [{"ref": c1}, {"ref": t}], "add")
            c2 = b.conv(y1, 8, 8, 3)
            return b.finish(["x"], [b.call("add", [{"ref": y1}, {"ref": c2}], "add")])
        raise AssertionError(case)
    x = np.random.default_rng(5).standard_normal((2, 8, 6, 8)).astype(np.float32)
    for case, fused in (("two readers", 0), ("residual later", 0), ("broadcast", 0), ("both orders", 2)):
        plan, weights = build(case)
        ctx = EmuCtx()
        want, r0 = run_plan(plan, weights, ctx, EmuK(), {"x": EmuK._put(x, ctx.buf(), None)}, record=True)
        folded = fold_channel_views(plan, r0.shapes)
        assert folded["folded"]["residual_adds_fused"] == fused, (case, folded["folded"])
        k = EmuK()
        got, _ = run_plan(folded, weights, ctx, k, {"x": EmuK._put(x, ctx.buf(), None)})
        assert getattr(k, "res_calls", 0) == fused and all(np.array_equal(p, q) for p, q in zip(want, got)), case
        if fused:
            acts = [st["args"][8]["int"] for st in folded["statements"] if st.get("fn") == "conv2d_res"]
            assert acts == [0, 2]


def detection_tail_plan(seed=4):
    """three levels [N, 10, h, w] (each a Concat along C of a 4-, a 3- and a 3-channel convolution) -> reshape to [N, 10, P_l] ->
    Concat along positions -> Transpose(0, 2, 1) -> Split into heads of 4 / 3 / 3 -> sigmoid of one head; two heads are outputs"""
    b = PlanBuilder(seed)
    levels = []
    feats = ["x", None, None]
    feats[1] = b.conv("x", 8, 8, 3, 2)
    feats[2] = b.conv(feats[1], 8, 8, 3, 2)
    for f, (h, w) in zip(feats, ((8, 12), (4, 6), (2, 3))):
        parts = [b.conv(f, 8, c, 1, silu=False) for c in (4, 3, 3)]
        cat = b.call("concat", [{"list": [{"ref": p} for p in parts]}, {"int": 1}], "cat")
        levels.append(b.call("reshape", [{"ref": cat}, b.ints([2, 10, h * w])], "lvl"))
        b.sts[-1]["bufs"] = 0
    pred = b.call("concat", [{"list": [{"ref": v} for v in levels]}, {"int": 2}], "pred")
    predt = b.call("transpose", [{"ref": pred}, b.ints([0, 2, 1])], "predt")
    box, cls, coef = b.call("split", [{"ref": predt}, {"int": 2}, b.ints([4, 3, 3])], "heads", 3)
    prob = b.call("sigmoid", [{"ref": cls}], "prob")
    return b.finish(["x"], [box, prob, coef])


def test_detection_tail_becomes_transposing_copies():
    plan, weights = detection_tail_plan()
    x = np.random.default_rng(8).standard_normal((2, 8, 8, 12)).astype(np.float32)
    ctx = EmuCtx()
    want, r0 = run_plan(plan, weights, ctx, EmuK(), {"x": EmuK._put(x, ctx.buf(), None)}, record=True)
    folded = fold_channel_views(plan, r0.shapes)
    assert folded["folded"]["transposed_splits_folded"] == 1 and folded["folded"]["concats_in_place"] == 3, folded["folded"]
    fns = [st.get("fn") for st in folded["statements"] if st["op"] == "call"]
    assert fns.count("transpose_cp") == 9 and "transpose" not in fns and "split" not in fns and fns.count("concat") == 0
    k = EmuK()
    got, _ = run_plan(folded, weights, ctx, k, {"x": EmuK._put(x, ctx.buf(), None)})
    assert k.tcp_calls == 9 and all(np.array_equal(p, q) for p, q in zip(want, got))
    # a second reader of the transposed tensor (or of the concatenation) keeps the three passes
    plan2, w2 = detection_tail_plan()
    t = [st for st in plan2["statements"] if st.get("fn") == "transpose"][0]["out"][0]
    plan2["statements"].append({"op": "call", "out": ["extra"], "fn": "sigmoid", "args": [{"ref": t}], "bufs": 1})
    plan2["outputs"].append("extra")
    from lele_amd.compiler.lower import allocate
    plan2["slots"] = allocate(plan2["statements"], plan2["outputs"])
    _w, r2 = run_plan(plan2, w2, ctx, EmuK(), {"x": EmuK._put(x, ctx.buf(), None)}, record=True)
    assert fold_channel_views(plan2, r2.shapes)["folded"]["transposed_splits_folded"] == 0


def rank3_tail_plan(sizes, seed=6):
    """lele's own Yolo26n-seg tail: rank-3 heads [N, c, P] -> Concat along C -> Transpose(0, 2, 1) -> Split along the last axis"""
    b = PlanBuilder(seed)
    heads = []
    for c, act in ((4, None), (3, "sigmoid"), (3, None)):
        v = b.call("reshape", [{"ref": b.conv("x", 8, c, 1, silu=False)}, b.ints([2, c, 96])], "flat")
        b.sts[-1]["bufs"] = 0
        heads.append(b.call(act, [{"ref": v}], "act") if act else v)
    u = b.call("concat", [{"list": [{"ref": h} for h in heads]}, {"int": 1}], "pred")
    t = b.call("transpose", [{"ref": u}, b.ints([0, 2, 1])], "predt")
    outs = b.call("split", [{"ref": t}, {"int": 2}, b.ints(sizes)], "heads", len(sizes))
    return b.finish(["x"], list(outs))


def test_rank3_tail_transpose_then_split():
    x = np.random.default_rng(9).standard_normal((2, 8, 8, 12)).astype(np.float32)
    for sizes, calls, concats in (([4, 3, 3], 3, 0), ([5, 5], 2, 1)):   # heads = the Concat's operands / cut elsewhere
        plan, weights = rank3_tail_plan(sizes)
        ctx = EmuCtx()
        want, r0 = run_plan(plan, weights, ctx, EmuK(), {"x": EmuK._put(x, ctx.buf(), None)}, record=True)
        folded = fold_channel_views(plan, r0.shapes)
        fns = [st.get("fn") for st in folded["statements"] if st["op"] == "call"]
        assert folded["folded"]["transposed_splits_folded"] == 1 and fns.count("transpose_cp") == calls and fns.count("concat") == concats \
            and "transpose" not in fns and "split" not in fns, (sizes, fns)
        k = EmuK()
        got, _ = run_plan(folded, weights, ctx, k, {"x": EmuK._put(x, ctx.buf(), None)})
        assert k.tcp_calls == calls and all(np.array_equal(p, q) for p, q in zip(want, got))


# --------------------------------------------------------------------------------------------- GPU
RES_CASES = [
    # n, c, h, w, oc, k, stride, group, act: every route of run_conv2d once
    (32, 32, 40, 44, 128, 3, 1, 1, 2),      # window kernel, 16-byte stores
    (40, 16, 30, 37, 64, 3, 1, 1, 1),       # window kernel, scalar stores (ow % 4 != 0)
    (32, 32, 80, 80, 64, 3, 2, 1, 2),       # stride-2 window kernel
    (32, 48, 80, 80, 64, 1, 1, 1, 2),       # 1 x 1 on the window kernel
    (64, 32, 40, 40, 128, 1, 1, 1, 2),      # 1 x 1 on the window kernel, 128-channel blocks
    (8, 96, 12, 12, 200, 3, 1, 1, 0),       # tiled GEMM, 16-byte stores
    (4, 24, 9, 7, 40, 3, 1, 1, 2),          # tiled / small GEMM, scalar stores
    (64, 130, 20, 20, 200, 1, 1, 1, 2),     # pointwise tiled GEMM
    (48, 16, 48, 48, 8, 3, 1, 1, 2),        # direct small-channel kernel
    (2, 128, 17, 17, 128, 3, 1, 128, 2),    # depthwise (+ the separate add)
    (1, 16, 33, 29, 24, 3, 1, 2, 1),        # grouped
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", RES_CASES, ids=[str(i) for i in range(len(RES_CASES))])
def test_conv2d_res_equals_convolution_then_add(ctx, case):
    """lele_hip_conv2d_res against the two calls it stands for (lele_hip_conv2d, then the Add): the same bits on every route, dense
    and with the input, the residual and the result as channel views"""
    from lele_amd import kernels as K
    n, c, h, w_, oc, k, s, g, act = case
    rng = np.random.default_rng(sum(case))
    x = rng.standard_normal((n, c, h, w_)).astype(np.float32)
    wt = (rng.standard_normal((oc, c // g, k, k)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(oc).astype(np.float32)
    fn = {0: K.conv2d, 1: lambda *a, **kw: K.conv2d_fused(*a, relu=True, **kw), 2: K.conv2d_silu}[act]
    pads, strides = [k // 2] * 4, [s, s]
    conv = fn(x, wt, bias, [1, 1], g, pads, strides, ctx=ctx).numpy()
    res = rng.standard_normal(conv.shape).astype(np.float32)
    want = K.add(conv, res, ctx=ctx).numpy()
    got = K.conv2d_res(x, wt, bias, res, [1, 1], g, pads, strides, act, ctx=ctx).numpy()
    assert np.array_equal(got, want)
    assert np.array_equal(K.conv2d_res(x, wt, None, res, [1, 1], g, pads, strides, act, ctx=ctx).numpy(),
                          K.add(fn(x, wt, None, [1, 1], g, pads, strides, ctx=ctx).numpy(), res, ctx=ctx).numpy())
    with pytest.raises(_lib.LeleError):
        K.conv2d_res(x, wt, bias, res[:, :, :-1], [1, 1], g, pads, strides, act, ctx=ctx)
    if g != 1:
        return
    # views: x = channels [3, 3 + c) of a wider tensor, res = channels [2, 2 + oc) of another, the result a window of a third
    oh, ow = conv.shape[2:]
    wide_x = rng.standard_normal((n, c + 5, h, w_)).astype(np.float32)
    wide_x[:, 3:3 + c] = x
    wide_r = rng.standard_normal((n, oc + 4, oh, ow)).astype(np.float32)
    wide_r[:, 2:2 + oc] = res
    xv = TensorView(ctx.buf().upload(wide_x)).channels(3, 3 + c)
    rv = TensorView(ctx.buf().upload(wide_r)).channels(2, 2 + oc)
    tot = oc + 3
    big = ctx.buf()
    big.upload(np.full((n, tot, oh, ow), -3.25, np.float32))
    out = K.conv2d_res(xv, wt, bias, rv, [1, 1], 1, pads, strides, act, out=big, out_window=(1 * oh * ow, tot * oh * ow), ctx=ctx)
    assert out.is_view and np.array_equal(out.numpy(), want)
    whole = big.to_numpy((n, tot, oh, ow))
    assert np.all(whole[:, :1] == -3.25) and np.all(whole[:, 1 + oc:] == -3.25)


@pytest.mark.gpu
def test_pitched_kernels_vs_oracle(ctx):
    """every *_pitched entry point reading a window of a wider tensor and writing a window of another one, against the oracle on
    dense copies -- and bit for bit against the library's own dense call (same kernels, only the addressing differs)"""
    from lele_amd import kernels as K
    rng = np.random.default_rng(0)
    n, ct, h, w = 3, 40, 24, 32
    wide = rng.standard_normal((n, ct, h, w)).astype(np.float32)
    src = TensorView(ctx.buf().upload(wide))
    cases = [(16, 16, 16, 3, 1, "silu"), (8, 32, 64, 1, 1, "silu"), (0, 8, 12, 3, 2, None), (24, 16, 64, 3, 1, None), (4, 3, 16, 3, 2, "silu")]
    for c0, cin, cout, k, s, act in cases:
        wt = (rng.standard_normal((cout, cin, k, k)) * 0.2).astype(np.float32)
        bias = rng.standard_normal(cout).astype(np.float32)
        xin = src.channels(c0, c0 + cin)
        dense_x = np.ascontiguousarray(wide[:, c0:c0 + cin])
        assert np.array_equal(xin.numpy(), dense_x)
        fn = K.conv2d_silu if act == "silu" else K.conv2d
        ref = fn(dense_x, wt, bias, [1, 1], 1, [k // 2] * 4, [s, s], ctx=ctx).numpy()
        oh, ow = ref.shape[2:]
        tot = cout + 9
        big = ctx.buf()
        big.reserve(4 * n * tot * oh * ow)
        sentinel = np.full((n, tot, oh, ow), 7.5, np.float32)
        big.upload(sentinel)
        got = fn(xin, wt, bias, [1, 1], 1, [k // 2] * 4, [s, s], out=big, out_window=(5 * oh * ow, tot * oh * ow), ctx=ctx)
        assert got.is_view and np.array_equal(got.numpy(), ref), (c0, cin, cout, k, s)
        whole = big.to_numpy((n, tot, oh, ow))
        assert np.array_equal(whole[:, 5:5 + cout], ref) and np.all(whole[:, :5] == 7.5) and np.all(whole[:, 5 + cout:] == 7.5)
        want = O.conv2d(dense_x, wt, bias, [1, 1], 1, [k // 2] * 4, [s, s], act)
        den = 1e-4 * np.maximum(np.abs(want), float(np.sqrt(np.mean(np.square(want, dtype=np.float64))))) + 1e-7
        assert float((np.abs(ref - want) / den).max()) <= 1.0
    # add / max_pool / resize / copy between windows
    a, b = src.channels(3, 19), src.channels(20, 36)
    big = ctx.buf()
    big.reserve(4 * n * 24 * h * w)
    big.upload(np.zeros((n, 24, h, w), np.float32))
    r = K.add(a, b, out=big, out_window=(8 * h * w, 24 * h * w), ctx=ctx)
    assert np.array_equal(r.numpy(), wide[:, 3:19] + wide[:, 20:36])
    assert np.array_equal(K.mul(a, wide[:, 20:36].copy(), ctx=ctx).numpy(), wide[:, 3:19] * wide[:, 20:36])
    # max-pool: `if val > max_val` (conv2d.rs:1230-1247) -- a NaN never wins and the first of equal values stays, so a window that
    # holds +0 then -0 gives +0.  The numpy reference (np.maximum) propagates NaN: compared on clean data; the special values are
    # checked by hand where they sit, and pitched == dense on everything.
    wz = wide.copy()
    wz[0, 5, 3, 3:5] = [0.0, -0.0]
    wz[0, 5, 1:6, 1:7] = np.minimum(wz[0, 5, 1:6, 1:7], 0.0) - (wz[0, 5, 1:6, 1:7] != 0) * 1.0   # everything else around them negative
    wz[0, 5, 3, 3:5] = [0.0, -0.0]
    wz[1, 6, 0, 0] = np.nan
    srcz = TensorView(ctx.buf().upload(wz))
    for kk, ss, pp in (([5, 5], [1, 1], [2, 2, 2, 2]), ([3, 3], [2, 2], [1, 1, 1, 1]), ([2, 2], [2, 2], [0, 0, 0, 0])):
        want = npref.max_pool2d(wide[:, 4:20], kk, ss, pp)
        assert np.array_equal(K.max_pool2d(src.channels(4, 20), kk, ss, pp, ctx=ctx).numpy().view(np.uint32), want.view(np.uint32)), kk
        dense = K.max_pool2d(np.ascontiguousarray(wz[:, 4:20]), kk, ss, pp, ctx=ctx).numpy()
        got = K.max_pool2d(srcz.channels(4, 20), kk, ss, pp, ctx=ctx).numpy()
        assert np.array_equal(got.view(np.uint32), dense.view(np.uint32)), kk
        ob = ctx.buf()
        ob.reserve(4 * n * 20 * want.shape[2] * want.shape[3])
        got = K.max_pool2d(srcz.channels(4, 20), kk, ss, pp, out=ob, out_window=(2 * want.shape[2] * want.shape[3], 20 * want.shape[2] * want.shape[3]), ctx=ctx)
        assert np.array_equal(got.numpy().view(np.uint32), dense.view(np.uint32))
        if kk == [5, 5]:
            assert not np.isnan(dense[1, 2]).any()                                   # channel 6 of the tensor = 2 of the window: the NaN lost
            assert dense[0, 1, 3, 3] == 0.0 and not np.signbit(dense[0, 1, 3, 3])    # +0 (first in scan order) beats -0
    up = K.resize_nearest(src.channels(7, 15), scales=[1, 1, 2, 2], ctx=ctx).numpy()
    assert np.array_equal(up, npref.resize_nearest(wide[:, 7:15], 2 * h, 2 * w))
    cp = K.copy_view(src.channels(30, 40), ctx=ctx)
    assert not cp.is_view and np.array_equal(cp.numpy(), wide[:, 30:40])
    ids = np.arange(n * 6 * 4, dtype=np.int64).reshape(n, 6, 4)
    assert np.array_equal(K.copy_view(TensorView(ctx.buf().upload(ids)).channels(1, 4), ctx=ctx).numpy(), ids[:, 1:4])
    # an operator without a pitched form refuses a view instead of reading it as if it were dense
    with pytest.raises(_lib.LeleError):
        K.sigmoid(src.channels(0, 4), ctx=ctx)
    with pytest.raises(_lib.LeleError):
        K.conv2d(src.channels(0, 8), np.zeros((8, 1, 3, 3), np.float32), None, [1, 1], 8, [1] * 4, [1, 1], ctx=ctx)


@pytest.mark.gpu
def test_transposing_copy_and_the_detection_tail(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(12)
    wide = rng.standard_normal((3, 116, 7, 9)).astype(np.float32)
    src = TensorView(ctx.buf().upload(wide))
    for c0, c1 in ((0, 4), (4, 84), (84, 116), (0, 116)):
        want = np.ascontiguousarray(wide[:, c0:c1].reshape(3, c1 - c0, 63).transpose(0, 2, 1))
        assert np.array_equal(K.transpose_cp(src.channels(c0, c1), ctx=ctx).numpy(), want)
        big = ctx.buf()
        big.upload(np.full((3, 100, c1 - c0), 2.5, np.float32))
        got = K.transpose_cp(src.channels(c0, c1), out=big, out_window=(20 * (c1 - c0), 100 * (c1 - c0)), ctx=ctx)
        whole = big.to_numpy((3, 100, c1 - c0))
        assert np.array_equal(got.numpy(), want) and np.array_equal(whole[:, 20:83], want) and np.all(whole[:, :20] == 2.5) and np.all(whole[:, 83:] == 2.5)
    plan, weights = detection_tail_plan()
    x = rng.standard_normal((2, 8, 8, 12)).astype(np.float32)
    xb = ctx.buf().upload(x)
    want, r0 = run_plan(plan, weights, ctx, None, {"x": TensorView(xb)}, record=True)
    folded = fold_channel_views(plan, r0.shapes)
    assert folded["folded"]["transposed_splits_folded"] == 1
    got = [o.numpy().copy() for o in Runner(folded, weights, ctx).run({"x": TensorView(xb)})]
    assert all(np.array_equal(a, b) for a, b in zip(want, got))


@pytest.mark.gpu
def test_folded_plan_equals_the_plan_it_came_from(ctx):
    plan, weights = c3k2_sppf_fpn_plan()
    for shape in ((3, 16, 16, 24), (1, 16, 40, 40), (5, 16, 8, 12)):
        x = np.random.default_rng(shape[0]).standard_normal(shape).astype(np.float32)
        xb = ctx.buf().upload(x)
        want, r0 = run_plan(plan, weights, ctx, None, {"x": TensorView(xb)}, record=True)
        folded = fold_channel_views(plan, r0.shapes)
        assert folded["folded"]["concats_in_place"] == 3
        r = Runner(folded, weights, ctx)
        got = [o.numpy().copy() for o in r.run({"x": TensorView(xb)})]
        for a, b in zip(want, got):
            assert np.array_equal(a, b)
        # recorded into a graph and replayed
        ctx.sync()
        ctx.graph_begin()
        outs = r.run({"x": TensorView(xb)})
        g = ctx.graph_end()
        for _ in range(2):
            g.launch()
        ctx.sync()
        assert all(np.array_equal(a, o.numpy()) for a, o in zip(want, outs))
        g.close()
        # against the emulation's oracle arithmetic
        ek = EmuCtx()
        ref, _ = run_plan(plan, weights, ek, EmuK(), {"x": EmuK._put(x, ek.buf(), None)})
        for a, b in zip(want, ref):
            den = 1e-4 * np.maximum(np.abs(b), float(np.sqrt(np.mean(np.square(b, dtype=np.float64))))) + 1e-7
            assert float((np.abs(a - b) / den).max()) <= 1.0


@pytest.mark.gpu
def test_yolo_shaped_graph_folded_equals_unfolded(ctx):
    """the Yolo26n-seg-shaped network of tools/yolo_graph.py at a small size and batch 3: every Concat / Split along C that can be a
    view is one, and the outputs are the unfolded plan's, bit for bit"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from yolo_graph import yolo_onnx
    from lele_amd.compiler import compile_model
    from lele_amd.plan import load_weights_bin
    data, _info = yolo_onnx(3, 128)
    plan, blob = compile_model(data, "yolo_small")
    weights = load_weights_bin(plan, blob)
    x = np.random.default_rng(0).uniform(0, 1, (3, 3, 128, 128)).astype(np.float32)
    xb = ctx.buf().upload(x)
    want, r0 = run_plan(plan, weights, ctx, None, {"images": TensorView(xb)}, record=True)
    folded = fold_channel_views(plan, r0.shapes)
    info = folded["folded"]
    assert info["concats_in_place"] >= 15 and info["splits_as_views"] >= 8, info
    got, r1 = run_plan(folded, weights, ctx, None, {"images": TensorView(xb)})
    assert r1.calls < r0.calls
    for a, b in zip(want, got):
        assert np.array_equal(a, b)


def test_a_concat_of_a_view_splits_results_is_never_left_as_a_plain_concat():
    """y = relu(x); a, b = split(y); z = concat([a, b]).  The Split's results become channel views because the Concat counts
    as a reader that copes with views; merging them back into y made every entry of the Concat dense and nothing could be written in
    place, so the Concat used to be dropped from the plan of in-place Concats -- and a plain `concat` was left reading views, which
    fails at run time.  No GPU: the structure of the folded plan."""
    from lele_amd.plan import fold_channel_views
    L = lambda *v: {"list": [{"int": int(i)} for i in v]}   # noqa: E731
    st = [{"op": "call", "out": ["y"], "fn": "relu", "args": [{"ref": "x"}], "bufs": 1},
          {"op": "call", "out": ["a", "b"], "fn": "split", "args": [{"ref": "y"}, {"int": 1}, L(3, 5)], "bufs": 2},
          {"op": "call", "out": ["z"], "fn": "concat", "args": [{"list": [{"ref": "a"}, {"ref": "b"}]}, {"int": 1}], "bufs": 1},
          {"op": "call", "out": ["w"], "fn": "relu", "args": [{"ref": "z"}], "bufs": 1}]
    plan = {"source": "t", "format": "lele_amd.plan/2", "inputs": ["x"], "outputs": ["w"], "slots": [], "statements": st, "weights": {}}
    shapes = {"x": [2, 8, 4, 4], "y": [2, 8, 4, 4], "a": [2, 3, 4, 4], "b": [2, 5, 4, 4], "z": [2, 8, 4, 4], "w": [2, 8, 4, 4]}
    out = fold_channel_views(plan, shapes)["statements"]
    views = {o for s_ in out if s_["op"] == "chview" for o in s_["out"]}
    for s_ in out:
        if s_["op"] == "call" and s_.get("fn") == "concat":
            ops = [v["ref"] for v in s_["args"][0]["list"]]
            assert not (set(ops) & views), "a plain concat reads channel views: %s" % ops
    if {"a", "b"} <= views:      # the Split became views: then z is a view of a reserved buffer y was copied into
        assert any(s_["op"] == "reserve" for s_ in out) and any(s_.get("fn") == "copy_view" for s_ in out)
        assert "z" in views
