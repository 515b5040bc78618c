"""ONNX graph -> device plan (the compiler back-end).

lele's compiler (src/compiler/mod.rs:311-373) turns an ONNX graph into Rust source: one `lele::kernels::<op>(...)`
statement per node, fused forms where a pattern of consecutive nodes matches (patterns.rs), output buffers assigned by a
liveness scan (mod.rs:148-290) and all constants packed into `<model>_weights.bin` (mod.rs:1381-1505).  This module does
the same job for the device: the product is a PLAN -- the statement list `lele_amd.plan.Runner` executes through the
C ABI and records into a hipGraph -- plus a weights file in lele's own format (raw little-endian tensors, 16-byte
aligned, identical contents stored once), so a plan lifted from lele-generated Rust (tools/lift_generated.py) and a
plan compiled here are interchangeable.

Differences from the Rust emitter, by design:
  * integer side computations (shapes, axes, slice bounds) are evaluated on the HOST -- at compile time when their
    operands are constant, otherwise by the runner without touching the device (hostops.py);
  * a pattern is fused only when its intermediate values have no other consumer (the reference does not check).
"""
import numpy as np

from . import hostops
from . import onnx_pb as pb

KIND_OF_DTYPE = {pb.FLOAT: "weight_f32", pb.UINT8: "weight_u8", pb.INT8: "weight_i8", pb.INT32: "weight_i32",
                 pb.INT64: "weight_i64", pb.FLOAT16: "weight_f16", pb.DOUBLE: "weight_f64", pb.BOOL: "weight_u8"}

UNARY = {"Relu": "relu", "Sigmoid": "sigmoid", "Tanh": "tanh_kernel", "Exp": "exp", "Log": "log", "Sqrt": "sqrt", "Neg": "neg",
         "Reciprocal": "reciprocal", "Erf": "erf", "Softplus": "softplus", "Sin": "sin", "Cos": "cos", "Not": "not_",
         "Abs": "abs", "Floor": "floor", "Ceil": "ceil"}
BINARY = {"Add": "add", "Sub": "sub", "Mul": "mul", "Div": "div", "Pow": "pow", "Equal": "equal", "Less": "less",
          "Greater": "greater", "PRelu": "prelu", "Mod": "mod_f32", "And": "and_", "Or": "or_"}
VIEW_OPS = {"Reshape", "Flatten", "Squeeze", "Unsqueeze", "Identity"}  # share the input's buffer (shape.rs:2-52)


class CompileError(Exception):
    pass


def sanitize(name):
    """src/compiler/mod.rs:1359-1370"""
    s = name.replace(".", "_").replace("/", "_").replace("-", "_").replace(":", "_")
    return "_" + s if s[:1].isdigit() else s


class WeightPacker:
    """`<model>_weights.bin`: src/compiler/mod.rs:1381-1505 (align 16, identical byte strings stored once)"""

    def __init__(self):
        self.blob = bytearray()
        self.by_content = {}

    def add(self, array, onnx_dtype):
        a = np.ascontiguousarray(array)
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        key = (raw, )  # byte-identical tensors share storage whatever their shape or dtype
        if key not in self.by_content:
            pad = (-len(self.blob)) % 16
            self.blob += b"\0" * pad
            self.by_content[key] = len(self.blob)
            self.blob += raw
        return [KIND_OF_DTYPE[onnx_dtype], self.by_content[key], len(raw), [int(d) for d in a.shape]]


def _attrs(node):
    out = {}
    for a in node.attribute:
        if a.t is not None:
            out[a.name] = a.t.array
        elif a.ints:
            out[a.name] = [int(v) for v in a.ints]
        elif a.floats:
            out[a.name] = [float(v) for v in a.floats]
        elif a.s is not None:
            out[a.name] = a.s.decode()
        elif a.i is not None:
            out[a.name] = int(a.i)
        elif a.f is not None:
            out[a.name] = float(a.f)
        elif a.g is not None:
            out[a.name] = a.g   # control flow: only `If` is lowered (as in lele: ops/control_flow.rs)
        else:
            out[a.name] = []
    return out


def captured_names(graph):
    """outer-scope values a sub-graph reads (its nodes' operands and its outputs, nested sub-graphs included, minus
    what the sub-graph defines itself), in first-use order"""
    defined = {t.name for t in graph.initializer} | {v.name for v in graph.input}
    for n in graph.node:
        defined.update(o for o in n.output if o)
    seen, out = set(), []

    def want(name):
        if name and name not in defined and name not in seen:
            seen.add(name)
            out.append(name)
    for n in graph.node:
        for i in n.input:
            want(i)
        for a in n.attribute:
            if a.g is not None:
                for c in captured_names(a.g):
                    want(c)
    for v in graph.output:
        want(v.name)
    return out


class Lowering:
    def __init__(self, model, name="model", extra_fusions=True, bind=None, graph=None, parent=None):
        """bind: {input name: value} -- graph inputs fixed at compile time (they become constants and fold; e.g. Silero's
        `sr`, whose `If` then inlines the taken branch).  graph/parent: a sub-graph lowered in its parent's scope."""
        self.model, self.name, self.extra_fusions = model, name, extra_fusions
        if parent is not None:
            self.consts, self.const_dtype = dict(parent.consts), dict(parent.const_dtype)
            for t in graph.initializer:
                self.consts[t.name], self.const_dtype[t.name] = t.array, t.data_type
            self.inputs, self.outputs = [], [v.name for v in graph.output]
            self.place, self.static_shape = dict(parent.place), dict(parent.static_shape)
            self.packer, self.weight_cache = parent.packer, parent.weight_cache
            self.statements, self.alias = [], {}
            self.branch_slots, self.if_count = parent.branch_slots, parent.if_count
            return
        self.branch_slots, self.if_count = [], [0]   # shared with the sub-graph lowerings: slots private to `If` branches
        g = model.graph
        self.consts = {}       # name -> numpy array (initializers, Constant nodes, folded results)
        self.const_dtype = {}  # name -> ONNX data type of the stored tensor
        for t in g.initializer:
            self.consts[t.name], self.const_dtype[t.name] = t.array, t.data_type
        for k, v in (bind or {}).items():
            if k not in {vi.name for vi in g.input}:
                raise CompileError("bind: %r is not a graph input" % k)
            self.add_const(k, np.asarray(v))
        self.inputs = [v for v in g.input if v.name not in self.consts]
        self.outputs = [v.name for v in g.output]
        self.place = {}        # name -> "host" | "dev" for run-time values
        for v in self.inputs:
            self.place[v.name] = "host" if v.elem_type in (pb.INT64, pb.INT32, pb.BOOL) else "dev"
        self.static_shape = {v.name: v.shape for v in self.inputs if v.shape and all(isinstance(d, int) and d > 0 for d in v.shape)}
        self.packer = WeightPacker()
        self.weight_cache = {}
        self.statements = []
        self.alias = {}        # view output -> the value whose buffer it shares

    # ---------------------------------------------------------------------------------------- constants
    def add_const(self, name, arr):
        arr = np.asarray(arr)
        self.consts[name] = arr
        if arr.dtype in pb.ONNX_OF:
            self.const_dtype[name] = pb.ONNX_OF[arr.dtype]
        else:
            self.consts[name] = arr.astype(np.int64 if arr.dtype.kind in "iub" else np.float32)
            self.const_dtype[name] = pb.INT64 if arr.dtype.kind in "iub" else pb.FLOAT

    def fold(self, nodes):
        """Constant nodes become constants; nodes whose operands are all constant are evaluated now (mod.rs:375-760)"""
        rest = []
        for n in nodes:
            at = _attrs(n)
            if n.op_type == "Constant":
                for key in ("value", "value_float", "value_int", "value_ints", "value_floats"):
                    if key in at:
                        v = at[key]
                        arr = v if isinstance(v, np.ndarray) else np.asarray(v, np.int64 if "int" in key else np.float32)
                        self.add_const(n.output[0], arr)
                        break
                else:
                    raise CompileError("Constant node %r without a supported value attribute" % n.name)
                continue
            if n.op_type == "Shape" and n.input[0] in self.static_shape:
                self.add_const(n.output[0], np.array(self.static_shape[n.input[0]], np.int64))
                continue
            if n.op_type == "If":
                if n.input[0] in self.consts:   # the condition is known now: the taken branch is inlined, the other one dropped
                    g = at["then_branch"] if bool(np.asarray(self.consts[n.input[0]]).reshape(-1)[0] != 0) else at["else_branch"]
                    for t in g.initializer:
                        self.consts[t.name], self.const_dtype[t.name] = t.array, t.data_type
                    rest += self.fold(list(g.node))
                    for bo, o in zip(g.output, n.output):
                        if bo.name in self.consts:
                            self.consts[o], self.const_dtype[o] = self.consts[bo.name], self.const_dtype[bo.name]
                        else:
                            rest.append(pb.Node("Identity", [bo.name], [o]))
                    continue
                # decided at run time: the values its branches read from this scope become visible operands of the node, so
                # that fusion privacy, view folding and liveness see those reads
                extra = [c for g in (at["then_branch"], at["else_branch"]) for c in captured_names(g)]
                m = pb.Node("If", [n.input[0]] + [c for i, c in enumerate(extra) if c not in extra[:i] and c != n.input[0]], n.output, n.name)
                m.attribute = n.attribute
                rest.append(m)
                continue
            ins = [self.consts.get(i) if i else None for i in n.input]
            if n.input and all(i == "" or i in self.consts for i in n.input):
                if n.op_type == "ConstantOfShape" and "value" in at:
                    at = dict(at, value=np.asarray(at["value"]).reshape(-1)[0])
                res = hostops.evaluate(n.op_type, ins, at)
                if res is not None:
                    for o, r in zip(n.output, res):
                        self.add_const(o, r)
                    continue
            rest.append(n)
        return rest

    def weight(self, name, as_f32=False):
        """argument node for a constant used as a device tensor"""
        key = (name, as_f32)
        if key not in self.weight_cache:
            arr, dt = self.consts[name], self.const_dtype[name]
            # an i64 / i32 / f64 constant meeting f32 arithmetic is converted once, here; u8 / i8 / f16 weights stay as they
            # are in the file and are widened by the loader, as lele's weight_u8 / weight_i8 / weight_f16 accessors do
            if as_f32 and dt in (pb.INT64, pb.INT32, pb.DOUBLE, pb.BOOL):
                arr, dt = arr.astype(np.float32), pb.FLOAT
            self.weight_cache[key] = {"weight": self.packer.add(arr, dt)}
        return self.weight_cache[key]

    # ---------------------------------------------------------------------------------------- operands
    def tensor(self, name, as_f32=True):
        if name in self.consts:
            return self.weight(name, as_f32)
        return {"ref": sanitize(name)}

    def opt_tensor(self, node, i):
        return self.tensor(node.input[i]) if len(node.input) > i and node.input[i] else {"none": 1}

    def ints(self, node, i, attr=None, at=None, default=None):
        """an integer-list operand: from input i (constant -> literal, run-time host value -> fetched by the runner),
        else from the attribute, else the default"""
        if len(node.input) > i and node.input[i]:
            nm = node.input[i]
            if nm in self.consts:
                return {"list": [{"int": int(v)} for v in np.asarray(self.consts[nm]).reshape(-1)]}
            if self.place.get(nm) != "host":
                raise CompileError("%s %r: operand %d (%s) must be a host integer value" % (node.op_type, node.name, i, nm))
            return {"ints": sanitize(nm)}
        if attr is not None and at is not None and attr in at:
            return {"list": [{"int": int(v)} for v in at[attr]]}
        return {"list": [{"int": int(v)} for v in (default or [])]}

    # ---------------------------------------------------------------------------------------- emission
    def emit(self, outs, fn, args, n_bufs=1, view=False):
        st = {"op": "call", "out": [sanitize(o) for o in outs], "fn": fn, "args": args, "bufs": 0 if view else n_bufs}
        self.statements.append(st)
        for o in outs:
            self.place[o] = "dev"
        return st

    def emit_host(self, node, at):
        ins = []
        for i in node.input:
            if not i:
                ins.append(None)
            elif i in self.consts:
                ins.append({"const": np.asarray(self.consts[i]).tolist(), "dtype": "i64" if self.consts[i].dtype.kind in "iub" else "f32"})
            elif node.op_type in ("Shape", "Size") or self.place.get(i) == "host":
                ins.append({"ref": sanitize(i)})
            else:
                raise CompileError("host op %s %r reads the device value %s" % (node.op_type, node.name, i))
        at = {k: (np.asarray(v).reshape(-1)[0].item() if isinstance(v, np.ndarray) else v) for k, v in at.items()}
        self.statements.append({"op": "host", "out": [sanitize(o) for o in node.output], "onnx": node.op_type, "in": ins, "attrs": at})
        for o in node.output:
            self.place[o] = "host"

    # ---------------------------------------------------------------------------------------- patterns
    def uses(self, nodes):
        cnt = {}
        for n in nodes:
            for i in n.input:
                cnt[i] = cnt.get(i, 0) + 1
        for o in self.outputs:
            cnt[o] = cnt.get(o, 0) + 1
        return cnt

    def match(self, nodes, k, cnt):
        """patterns.rs: fused forms over CONSECUTIVE nodes -> (nodes consumed, emitter) or None"""
        def ops(*names):
            return k + len(names) <= len(nodes) and all(nodes[k + j].op_type == nm for j, nm in enumerate(names))

        def private(*vals):  # every intermediate is read exactly once (by the next node of the pattern)
            return all(cnt.get(v, 0) == 1 for v in vals)

        n = nodes[k:k + 9]
        # LayerNorm (patterns.rs:6-119): ReduceMean Sub Pow ReduceMean Add Sqrt Div Mul Add
        if ops("ReduceMean", "Sub", "Pow", "ReduceMean", "Add", "Sqrt", "Div", "Mul", "Add"):
            x, mean, sub, pw, var, ade, std, norm, scaled = (n[0].input[0], n[0].output[0], n[1].output[0], n[2].output[0],
                                                              n[3].output[0], n[4].output[0], n[5].output[0], n[6].output[0], n[7].output[0])
            ok = (x in n[1].input and mean in n[1].input and n[2].input[0] == sub and n[3].input[0] == pw and n[4].input[0] == var
                  and n[5].input[0] == ade and n[6].input[:2] == [sub, std] and n[7].input[0] == norm and scaled in n[8].input
                  and private(mean, pw, var, ade, std, norm, scaled) and cnt.get(sub, 0) == 2)
            eps_n, scale_n = n[4].input[1], n[7].input[1]
            bias_n = n[8].input[1] if n[8].input[0] == scaled else n[8].input[0]
            ax = _attrs(n[0]).get("axes", [-1])
            if ok and all(v in self.consts for v in (eps_n, scale_n, bias_n)) and ax in ([-1],):
                eps = float(np.asarray(self.consts[eps_n]).reshape(-1)[0])
                return 9, lambda: self.emit([n[8].output[0]], "layer_norm", [self.tensor(x), self.weight(scale_n, True), self.weight(bias_n, True),
                                                                             {"int": -1}, {"float": eps}])
        # Quantized Linear (+ ReLU) (patterns.rs:121-432): DynamicQuantizeLinear Mul MatMulInteger Cast Mul Add [Relu]
        if ops("DynamicQuantizeLinear", "Mul", "MatMulInteger", "Cast", "Mul", "Add"):
            q, s, z = n[0].output[:3]
            comb, mm, cast, deq = n[1].output[0], n[2].output[0], n[3].output[0], n[4].output[0]
            ok = (s in n[1].input and n[2].input[0] == q and len(n[2].input) >= 4 and n[2].input[2] == z and n[3].input[0] == mm
                  and cast in n[4].input and comb in n[4].input and deq in n[5].input and private(q, s, z, comb, mm, cast, deq))
            ws_n = n[1].input[1] if n[1].input[0] == s else n[1].input[0]
            bias_n = n[5].input[1] if n[5].input[0] == deq else n[5].input[0]
            w_n, wz_n = n[2].input[1], n[2].input[3] if len(n[2].input) >= 4 else ""
            if ok and all(v in self.consts for v in (ws_n, bias_n, w_n, wz_n)):
                relu = ops("DynamicQuantizeLinear", "Mul", "MatMulInteger", "Cast", "Mul", "Add", "Relu") and \
                    nodes[k + 6].input[0] == n[5].output[0] and private(n[5].output[0])
                out = nodes[k + 6].output[0] if relu else n[5].output[0]
                args = [self.tensor(n[0].input[0]), self.weight(w_n, True), self.weight(ws_n, True), self.weight(wz_n, True),
                        self.weight(bias_n, True), {"bool": bool(relu)}]
                return (7 if relu else 6), lambda: self.emit([out], "fused_quantized_linear", args)
        # Conv + Sigmoid + Mul -> conv2d_silu (patterns.rs:704-840); Conv + Relu (559-703)
        if ops("Conv", "Sigmoid", "Mul") and len(_attrs(n[0]).get("kernel_shape", [1])) >= 2:
            c, sg = n[0].output[0], n[1].output[0]
            if n[1].input[0] == c and sorted(n[2].input) == sorted([c, sg]) and cnt.get(c, 0) == 2 and private(sg):
                return 3, lambda: self.conv(n[0], "conv2d_silu", n[2].output[0])
        if ops("Conv", "Relu") and n[1].input[0] == n[0].output[0] and private(n[0].output[0]):
            return 2, lambda: self.conv(n[0], None, n[1].output[0], relu=True)
        # Sigmoid + Mul -> silu (patterns.rs:1004-1062)
        if ops("Sigmoid", "Mul"):
            x, sg = n[0].input[0], n[0].output[0]
            if sorted(n[1].input) == sorted([x, sg]) and private(sg) and x not in self.consts:
                return 2, lambda: self.emit([n[1].output[0]], "silu", [self.tensor(x)])
        # MatMul + Add(bias) -> matmul_fused_add (patterns.rs:1063-1122); only for a constant 1-D bias
        if ops("MatMul", "Add") and n[0].output[0] in n[1].input and private(n[0].output[0]):
            bias_n = n[1].input[1] if n[1].input[0] == n[0].output[0] else n[1].input[0]
            if bias_n in self.consts and np.asarray(self.consts[bias_n]).ndim == 1 and self.consts[bias_n].dtype.kind == "f":
                return 2, lambda: self.emit([n[1].output[0]], "matmul_fused_add", [self.tensor(n[0].input[0]), self.tensor(n[0].input[1]),
                                                                                    self.weight(bias_n, True)])
        if not self.extra_fusions:
            return None
        # ---- fused forms beyond patterns.rs, each bit-identical to the sequence it replaces (include/lele_hip.h) ----
        # Mul by a one-element constant -> Softmax over the last axis (attention score scaling)
        if ops("Mul", "Softmax") and n[1].input[0] == n[0].output[0] and private(n[0].output[0]) and _attrs(n[1]).get("axis", -1) == -1:
            sc = [i for i in n[0].input if i in self.consts and self.consts[i].size == 1 and self.consts[i].dtype.kind == "f"]
            xs = [i for i in n[0].input if i not in self.consts]
            if len(sc) == 1 and len(xs) == 1:
                return 2, lambda: self.emit([n[1].output[0]], "softmax_scaled", [self.tensor(xs[0]), self.weight(sc[0], True), {"int": -1}])
        # Slice, Pow, Slice, Pow, Add, Sqrt on two ranges of one axis of the same tensor (the magnitude of a [re | im] spectrum)
        if ops("Slice", "Pow", "Slice", "Pow", "Add", "Sqrt"):
            def rng(sl):  # constant single-axis unit-step slice -> (axis, start, end)
                c = [self.consts.get(i) if i else None for i in sl.input[1:5]] + [None] * 4
                if c[0] is None or c[1] is None or c[2] is None or not (np.asarray(c[0]).size == np.asarray(c[1]).size == np.asarray(c[2]).size == 1):
                    return None
                if c[3] is not None and [int(v) for v in np.asarray(c[3]).reshape(-1)] != [1]:
                    return None
                return tuple(int(np.asarray(v).reshape(-1)[0]) for v in (c[2], c[0], c[1]))
            x, ra, rb = n[0].input[0], rng(n[0]), rng(n[2])
            e = [[i for i in p.input if i != s_.output[0]] for p, s_ in ((n[1], n[0]), (n[3], n[2]))]
            ok = (ra and rb and ra[0] == rb[0] and n[2].input[0] == x and x not in self.consts and n[1].input[0] == n[0].output[0]
                  and n[3].input[0] == n[2].output[0] and sorted(n[4].input) == sorted([n[1].output[0], n[3].output[0]])
                  and n[5].input[0] == n[4].output[0] and private(n[0].output[0], n[1].output[0], n[2].output[0], n[3].output[0], n[4].output[0])
                  and all(len(v) == 1 and v[0] in self.consts and self.consts[v[0]].size == 1 for v in e))
            if ok:
                first, second = (0, 1) if n[4].input[0] == n[1].output[0] else (1, 0)   # operand order of the Add: a + b == b + a bit for bit
                r, ex = (ra, rb), (e[0][0], e[1][0])
                pair = lambda q: {"list": [{"int": q[1]}, {"int": q[2]}]}  # noqa: E731
                return 6, lambda: self.emit([n[5].output[0]], "halves_pow_add_sqrt",
                                            [self.tensor(x), {"int": ra[0]}, pair(r[first]), pair(r[second]), self.weight(ex[first], True),
                                             self.weight(ex[second], True)])
        # Add -> Add: (a + b) + c in one pass (two residual connections in a row)
        if ops("Add", "Add") and n[0].output[0] in n[1].input and private(n[0].output[0]):
            c = [i for i in n[1].input if i != n[0].output[0]]
            if len(c) == 1 and not any(i in self.consts for i in list(n[0].input) + c):  # x + y == y + x bit for bit
                return 2, lambda: self.emit([n[1].output[0]], "add3", [self.tensor(n[0].input[0]), self.tensor(n[0].input[1]), self.tensor(c[0])])
        return None

    # ---------------------------------------------------------------------------------------- per-op lowering
    def conv(self, node, fn, out, relu=False):
        at = _attrs(node)
        wname = node.input[1]
        rank = np.asarray(self.consts[wname]).ndim if wname in self.consts else 3  # ops/nn.rs:46-54: unknown weights -> conv1d
        geom = [{"list": [{"int": v} for v in at.get("dilations", [])]}, {"int": at.get("group", 1)},
                {"list": [{"int": v} for v in at.get("pads", [])]}, {"list": [{"int": v} for v in at.get("strides", [])]}]
        if at.get("auto_pad", "NOTSET") not in ("NOTSET", ""):
            raise CompileError("Conv %r: auto_pad=%s is not supported (lele reads explicit pads only)" % (node.name, at["auto_pad"]))
        head = [self.tensor(node.input[0]), self.tensor(wname), self.opt_tensor(node, 2)]
        if fn is None:
            base = "conv2d" if rank >= 4 else "conv1d"
            fn = base + "_fused" if relu else base
        args = head + geom + ([{"bool": True}] if fn.endswith("_fused") else [])
        return self.emit([out], fn, args)

    def lower_node(self, node):
        op, at, I, O = node.op_type, _attrs(node), node.input, node.output
        T = self.tensor
        ilist = lambda key, d=None: {"list": [{"int": int(v)} for v in at.get(key, d or [])]}  # noqa: E731
        if op in UNARY:
            return self.emit(O, UNARY[op], [T(I[0])])
        if op in BINARY:
            return self.emit(O, BINARY[op], [T(I[0]), T(I[1])])
        if op in ("Max", "Min"):
            fn, acc = op.lower(), T(I[0])
            for j, nm in enumerate(I[1:]):
                last = j == len(I) - 2
                tmp = O[0] if last else "%s__%s%d" % (O[0], fn, j)
                self.emit([tmp], fn, [acc, T(nm)])
                acc = {"ref": sanitize(tmp)}
            return None
        if op == "MatMul":
            return self.emit(O, "matmul", [T(I[0]), T(I[1])])
        if op == "Gemm":
            return self.emit(O, "gemm", [T(I[0]), T(I[1]), self.opt_tensor(node, 2), {"float": at.get("alpha", 1.0)},
                                         {"float": at.get("beta", 1.0)}, {"bool": bool(at.get("transA", 0))}, {"bool": bool(at.get("transB", 0))}])
        if op == "Conv":
            return self.conv(node, None, O[0])
        if op == "ConvTranspose":
            if at.get("output_padding") and any(at["output_padding"]):
                raise CompileError("ConvTranspose %r: output_padding is not supported" % node.name)
            return self.emit(O, "conv_transpose", [T(I[0]), T(I[1]), self.opt_tensor(node, 2), ilist("dilations"), {"int": at.get("group", 1)},
                                                   ilist("pads"), ilist("strides")])
        if op == "ConvInteger":
            return self.emit(O, "conv_integer", [T(I[0]), T(I[1]), self.opt_tensor(node, 2), self.opt_tensor(node, 3), ilist("dilations"),
                                                 {"int": at.get("group", 1)}, ilist("pads"), ilist("strides")])
        if op == "MatMulInteger":
            return self.emit(O, "mat_mul_integer", [T(I[0]), T(I[1]), self.opt_tensor(node, 2), self.opt_tensor(node, 3)])
        if op == "DynamicQuantizeLinear":
            return self.emit(O, "dynamic_quantize_linear", [T(I[0])], n_bufs=3)
        if op == "LSTM":
            return self.emit([o for o in O] + ["%s__pad%d" % (O[0], j) for j in range(3 - len(O))], "lstm",
                             [T(I[0]), T(I[1]), T(I[2])] + [self.opt_tensor(node, j) for j in (3, 4, 5, 6)], n_bufs=3)
        if op == "GRU":
            return self.emit([o for o in O] + ["%s__pad%d" % (O[0], j) for j in range(2 - len(O))], "gru",
                             [T(I[0]), T(I[1]), T(I[2]), self.opt_tensor(node, 3), self.opt_tensor(node, 5),
                              {"bool": bool(at.get("linear_before_reset", 0))}], n_bufs=2)
        if op == "LayerNormalization":
            if len(I) < 3 or not I[1] or not I[2]:
                raise CompileError("LayerNormalization %r: scale and bias are required" % node.name)
            return self.emit(O[:1], "layer_norm", [T(I[0]), T(I[1]), T(I[2]), {"int": at.get("axis", -1)}, {"float": at.get("epsilon", 1e-5)}])
        if op == "BatchNormalization":
            return self.emit(O[:1], "batch_norm", [T(I[j]) for j in range(5)] + [{"float": at.get("epsilon", 1e-5)}])
        if op == "Softmax":
            return self.emit(O, "softmax", [T(I[0]), {"int": at.get("axis", -1)}])
        if op == "MaxPool":
            return self.emit(O[:1], "max_pool2d", [T(I[0]), ilist("kernel_shape"), ilist("strides"), ilist("pads"), ilist("dilations"),
                                                   {"bool": bool(at.get("ceil_mode", 0))}])
        if op == "Resize":
            mode = at.get("coordinate_transformation_mode", "half_pixel")
            if at.get("mode", "nearest") != "nearest":
                raise CompileError("Resize %r: only mode=nearest exists in lele (conv2d.rs:1261)" % node.name)
            if len(I) > 3 and I[3]:
                return self.emit(O, "resize_nearest", [T(I[0]), {"none": 1}, {"some": self.ints(node, 3)}, {"str": mode}])
            sc = I[2]
            if sc not in self.consts:
                raise CompileError("Resize %r: run-time scales are not supported" % node.name)
            return self.emit(O, "resize_nearest", [T(I[0]), {"some": {"list": [{"float": float(v)} for v in np.asarray(self.consts[sc]).reshape(-1)]}},
                                                   {"none": 1}, {"str": mode}])
        if op == "Transpose":
            st = self.emit(O, "transpose", [T(I[0]), ilist("perm")])
            # A permutation that only moves size-1 axes changes no byte: the runners then hand the input on as a view instead of
            # copying it (known only at run time, shapes being dynamic).  allocate() keeps the input's buffer alive as long as
            # the result for such statements, on top of giving the result a slot of its own.
            st["may_alias"] = True
            return st
        if op == "Reshape":
            return self.emit(O, "reshape", [T(I[0]), self.ints(node, 1)], view=True)
        if op == "Flatten":
            return self.emit(O, "flatten", [T(I[0]), {"int": at.get("axis", 1)}], view=True)
        if op in ("Unsqueeze", "Squeeze"):
            return self.emit(O, op.lower(), [T(I[0]), self.ints(node, 1, "axes", at)], view=True)
        if op == "Identity":
            return self.emit(O, "identity", [T(I[0])], view=True)
        if op == "Concat":
            return self.emit(O, "concat", [{"list": [T(nm) for nm in I]}, {"int": at.get("axis", 0)}])
        if op == "Where":
            return self.emit(O, "where_op", [T(I[0]), T(I[1]), T(I[2])])
        if op in ("Gather", "GatherElements"):
            return self.emit(O, "gather" if op == "Gather" else "gather_elements", [T(I[0]), T(I[1]), {"int": at.get("axis", 0)}])
        if op == "Slice":
            if len(I) == 1:  # opset < 10: attributes
                return self.emit(O, "slice", [T(I[0]), ilist("starts"), ilist("ends"), ilist("axes"), {"list": []}])
            return self.emit(O, "slice", [T(I[0]), self.ints(node, 1), self.ints(node, 2), self.ints(node, 3), self.ints(node, 4)])
        if op == "Expand":
            return self.emit(O, "expand", [T(I[0]), self.ints(node, 1)])
        if op == "Tile":
            return self.emit(O, "tile", [T(I[0]), self.ints(node, 1)])
        if op == "Split":
            if not (len(I) > 1 and I[1]) and "split" not in at:
                # the Rust emitter writes zeros here (ops/tensor.rs:330-338) and the kernel then panics on the size check
                raise CompileError("Split %r: no explicit sizes (equal split by output count) -- not supported by lele's split kernel" % node.name)
            return self.emit(O, "split", [T(I[0]), {"int": at.get("axis", 0)}, self.ints(node, 1, "split", at)], n_bufs=len(O))
        if op == "Pad":
            if len(I) > 1 and I[1]:
                pads = self.ints(node, 1)
            else:
                pads = ilist("pads")
            if len(I) > 2 and I[2]:  # lele::kernels::pad reads the fill value on the host (manipulation.rs:382): pass a literal
                cv = ({"array": [float(np.asarray(self.consts[I[2]]).reshape(-1)[0])], "dtype": "f32"} if I[2] in self.consts and self.consts[I[2]].size
                      else ({"none": 1} if I[2] in self.consts else self.tensor(I[2])))
            else:
                cv = {"array": [float(at["value"])], "dtype": "f32"} if "value" in at else {"none": 1}
            return self.emit(O, "pad", [T(I[0]), pads, cv, {"str": at.get("mode", "constant")}])
        if op in ("ReduceMean", "ReduceSum", "ReduceMax", "ReduceL2"):
            fn = {"ReduceMean": "reduce_mean", "ReduceSum": "reduce_sum", "ReduceMax": "reduce_max", "ReduceL2": "reduce_l2"}[op]
            return self.emit(O, fn, [T(I[0]), self.ints(node, 1, "axes", at), {"bool": bool(at.get("keepdims", 1))}])
        if op == "Clip":
            def bound(i, key):  # lele::kernels::clip reads the bounds on the host (math.rs:1984): pass literals
                if len(I) > i and I[i]:
                    if I[i] in self.consts:
                        return {"array": [float(np.asarray(self.consts[I[i]]).reshape(-1)[0])], "dtype": "f32"}
                    return self.tensor(I[i])
                return {"array": [float(at[key])], "dtype": "f32"} if key in at else {"none": 1}
            return self.emit(O, "clip", [T(I[0]), bound(1, "min"), bound(2, "max")])
        if op == "Cast":
            to = at.get("to", pb.FLOAT)
            if to in (pb.FLOAT, pb.FLOAT16, pb.DOUBLE):
                return self.emit(O, "identity", [T(I[0])], view=True)  # f32 -> f32: `clone()` upstream (ops/tensor.rs Cast)
            if to in (pb.INT64, pb.INT32, pb.BOOL):
                return self.emit(O, "cast_to_i64", [T(I[0])])
            raise CompileError("Cast %r: target type %d is not supported" % (node.name, to))
        if op == "TopK":
            k = self.ints(node, 1)
            return self.emit(O, "topk", [T(I[0]), {"first": k}, {"int": at.get("axis", -1)}, {"bool": bool(at.get("largest", 1))},
                                         {"bool": bool(at.get("sorted", 1))}], n_bufs=2)
        if op == "ConstantOfShape":
            val = float(np.asarray(at.get("value", np.float32(0.0))).reshape(-1)[0])
            return self.emit(O, "constant_of_shape", [self.ints(node, 0), {"float": val}])
        if op == "STFT":
            return self.emit(O, "stft", [T(I[0])] + [{"first": self.ints(node, j)} for j in (1,)] + [self.opt_tensor(node, 2)], n_bufs=1)
        if op == "_MatMulView":  # synthesised by fold_matmul_views
            opt = lambda v: {"none": 1} if v is None else {"list": [{"int": int(d)} for d in v]}  # noqa: E731
            return self.emit(O, "matmul_view", [T(I[0]), {"chain": node.a_chain}, T(I[1]), {"chain": node.b_chain}, opt(node.out_perm),
                                                opt(node.out_reshape)])
        if op == "_Tlc":  # synthesised by fold_time_major_conv
            return self.emit(O, "depthwise_conv1d_tlc", [T(I[0]), T(I[1]), self.opt_tensor(node, 2), {"int": node.pl}, {"int": node.pr}, {"bool": False},
                                                         {"int": node.x_offset}, {"bool": bool(node.add_input)}])
        if op == "_ViewCopy":  # synthesised by push_views
            return self.emit(O, "view_copy", [T(I[0]), {"chain": node.chain}])
        raise CompileError("ONNX operator %s (%r) is not supported by this back-end" % (op, node.name))

    # ---------------------------------------------------------------------------------------- view chains
    @staticmethod
    def _links(nodes):
        producer, consumers = {}, {}
        for idx, n in enumerate(nodes):
            for o in n.output:
                producer[o] = idx
            for i in n.input:
                consumers.setdefault(i, []).append(idx)
        return producer, consumers

    def fold_time_major_conv(self, nodes):
        """Transpose(0,2,1) -> depthwise Conv (k in 3, 5, 7, 11; stride 1) -> Transpose(0,2,1) [-> Add with the block's own input]:
        an FSMN memory block exported channel-major, computed on the time-major tensor it starts from (`depthwise_conv1d_tlc`)."""
        cnt = self.uses(nodes)
        producer, consumers = self._links(nodes)
        drop, replace = set(), {}
        for idx, c in enumerate(nodes):
            if c.op_type != "Conv" or c.input[0] not in producer or len(c.input) < 2 or c.input[1] not in self.consts:
                continue
            t0 = nodes[producer[c.input[0]]]
            use = consumers.get(c.output[0], [])
            if t0.op_type != "Transpose" or len(use) != 1 or nodes[use[0]].op_type != "Transpose":
                continue
            t1, a0, a1, a2 = nodes[use[0]], _attrs(t0), _attrs(c), _attrs(nodes[use[0]])
            w = self.consts[c.input[1]]
            bias_ok = len(c.input) < 3 or not c.input[2] or c.input[2] in self.consts
            if not (a0.get("perm") == [0, 2, 1] and a2.get("perm") == [0, 2, 1] and cnt.get(t0.output[0], 0) == 1 and cnt.get(c.output[0], 0) == 1
                    and w.ndim == 3 and w.shape[1] == 1 and a1.get("group", 1) == w.shape[0] and w.shape[2] in (3, 5, 7, 11) and bias_ok
                    and all(v == 1 for v in a1.get("strides", [1])) and all(v == 1 for v in a1.get("dilations", [1]))
                    and a1.get("auto_pad", "NOTSET") in ("NOTSET", "") and t0.input[0] not in self.consts):
                continue
            pads = a1.get("pads", [])
            v = pb.Node("_Tlc", [t0.input[0], c.input[1]] + ([c.input[2]] if len(c.input) > 2 and c.input[2] else []), [t1.output[0]])
            v.pl, v.pr = (pads[0] if len(pads) >= 1 else 0), (pads[1] if len(pads) >= 2 else 0)  # conv1d.rs:886-887
            v.x_offset, v.add_input, v.kw = 0, False, int(w.shape[2])
            # memory + input: Add(block output, block input) -- only when the lengths agree (pl + pr == k - 1)
            use2 = consumers.get(t1.output[0], [])
            if len(use2) == 1 and cnt.get(t1.output[0], 0) == 1 and nodes[use2[0]].op_type == "Add" and v.pl + v.pr == v.kw - 1:
                add = nodes[use2[0]]
                other = [i for i in add.input if i != t1.output[0]]
                if other == [t0.input[0]]:
                    v.add_input, v.output = True, [add.output[0]]
                    drop.add(use2[0])
            drop.update((producer[c.input[0]], idx))
            replace[use[0]] = [v]
        return [m for idx, n in enumerate(nodes) for m in (replace.get(idx, [n]) if idx not in drop or idx in replace else [])]

    def push_views(self, nodes, cnt):
        """Split -> Reshape -> Transpose (the head split of a packed QKV projection): every output that is read only through
        such a chain becomes ONE strided copy straight from the Split's input (`view_copy`), a time-major convolution reads
        its channel range of the packed tensor in place, and the Split shrinks to plain slices of the outputs that are still
        read directly.  Exact copies throughout, so nothing changes but the number of passes over the data."""
        nodes = self.fold_time_major_conv(nodes)
        cnt = self.uses(nodes)
        _producer, consumers = self._links(nodes)
        drop, replace = set(), {}
        for idx, n in enumerate(nodes):
            if n.op_type != "Split" or n.input[0] in self.consts:
                continue
            at = _attrs(n)
            sizes = self.consts.get(n.input[1]) if len(n.input) > 1 and n.input[1] else at.get("split")
            if sizes is None:
                continue
            sizes = [int(v) for v in np.asarray(sizes).reshape(-1)]
            if len(sizes) != len(n.output):
                continue
            axis, starts = at.get("axis", 0), [int(v) for v in np.cumsum([0] + sizes[:-1])]
            rewrites, direct = [], []
            for j, o in enumerate(n.output):
                pending, materialise = [], False
                for u in consumers.get(o, []):
                    un = nodes[u]
                    if un.op_type == "Reshape" and un.input[0] == o and un.input[1] in self.consts:
                        use2 = consumers.get(un.output[0], [])
                        if len(use2) == 1 and cnt.get(un.output[0], 0) == 1 and nodes[use2[0]].op_type == "Transpose" and _attrs(nodes[use2[0]]).get("perm"):
                            t = nodes[use2[0]]
                            pending.append(("chain", u, use2[0], [["slice", axis, starts[j], sizes[j]],
                                                                 ["reshape", [int(v) for v in np.asarray(self.consts[un.input[1]]).reshape(-1)]],
                                                                 ["transpose", _attrs(t)["perm"]]], t.output[0]))
                            continue
                    if un.op_type == "_Tlc" and un.input[0] == o and un.x_offset == 0 and axis in (-1, 2) and sizes[j] == int(self.consts[un.input[1]].shape[0]):
                        pending.append(("tlc", u, starts[j]))
                        continue
                    materialise = True
                if cnt.get(o, 0) > len(consumers.get(o, [])):   # also a graph output
                    materialise = True
                rewrites += pending
                if materialise and cnt.get(o, 0):
                    direct.append(j)
            if not rewrites:
                continue
            drop.add(idx)
            pre = []
            for j in direct:
                v = pb.Node("_ViewCopy", [n.input[0]], [n.output[j]])
                v.chain = [["slice", axis, starts[j], sizes[j]]]
                pre.append(v)
            replace[idx] = pre
            for rw in rewrites:
                if rw[0] == "chain":
                    _, ri, ti, chain, out = rw
                    drop.add(ri)
                    v = pb.Node("_ViewCopy", [n.input[0]], [out])
                    v.chain = chain
                    replace[ti] = [v]
                else:
                    _, ui, start = rw
                    nodes[ui].input[0], nodes[ui].x_offset = n.input[0], start
        out_nodes = []
        for idx, n in enumerate(nodes):
            if idx in replace:
                out_nodes += replace[idx]
            elif idx not in drop:
                out_nodes.append(n)
        return self.fold_matmul_views(out_nodes)

    def fold_matmul_views(self, nodes):
        """MatMul whose operands are private view copies (head views of a packed projection, a Reshape -> Transpose of a
        tensor) and/or whose result is only read through Transpose [-> Reshape]: the views go into the GEMM's loaders and
        store (`matmul_view`) -- same kernels and tiles, so the same bits, without the copies."""
        cnt = self.uses(nodes)
        producer, consumers = {}, {}
        for idx, n in enumerate(nodes):
            for o in n.output:
                producer[o] = idx
            for i in n.input:
                consumers.setdefault(i, []).append(idx)
        drop, replace = set(), {}

        def unit_dim_stays_inner(chain, rank_hint=None):
            # the source's innermost (unit-stride) dimension must end up among the last two logical dimensions
            pos = None
            for step in chain:
                if step[0] == "reshape":
                    pos = len(step[1]) - 1
                elif step[0] == "transpose":
                    perm = step[1]
                    src = (len(perm) - 1) if pos is None else pos
                    if src not in [p % len(perm) for p in perm[-2:]]:
                        return False
                    pos = [p % len(perm) for p in perm].index(src)
            return True

        def operand(name):
            """(source, chain, nodes to drop) for one MatMul operand"""
            i = producer.get(name)
            if i is None or cnt.get(name, 0) != 1 or i in drop:
                return name, [], []
            n = nodes[i]
            if n.op_type == "_ViewCopy" and unit_dim_stays_inner(n.chain):
                return n.input[0], n.chain, [i]
            if n.op_type == "Transpose" and _attrs(n).get("perm"):
                perm = _attrs(n)["perm"]
                j = producer.get(n.input[0])
                if j is not None and cnt.get(n.input[0], 0) == 1 and nodes[j].op_type == "Reshape" and nodes[j].input[1] in self.consts \
                        and nodes[j].input[0] not in self.consts:
                    chain = [["reshape", [int(v) for v in np.asarray(self.consts[nodes[j].input[1]]).reshape(-1)]], ["transpose", perm]]
                    if unit_dim_stays_inner(chain):
                        return nodes[j].input[0], chain, [i, j]
                chain = [["transpose", perm]]
                if n.input[0] not in self.consts and unit_dim_stays_inner(chain):
                    return n.input[0], chain, [i]
            return name, [], []

        for idx, n in enumerate(nodes):
            if n.op_type != "MatMul" or any(i in self.consts for i in n.input):
                continue
            a_src, a_chain, a_drop = operand(n.input[0])
            b_src, b_chain, b_drop = operand(n.input[1])
            out, out_perm, out_reshape, o_drop = n.output[0], None, None, []
            use = consumers.get(out, [])
            if len(use) == 1 and cnt.get(out, 0) == 1 and nodes[use[0]].op_type == "Transpose":
                perm = _attrs(nodes[use[0]]).get("perm")
                if perm and perm[-1] % len(perm) == len(perm) - 1:      # n stays innermost: the store stays row-contiguous
                    out_perm, o_drop, out = perm, [use[0]], nodes[use[0]].output[0]
                    use2 = consumers.get(out, [])
                    if len(use2) == 1 and cnt.get(out, 0) == 1 and nodes[use2[0]].op_type == "Reshape" and nodes[use2[0]].input[1] in self.consts:
                        out_reshape = [int(v) for v in np.asarray(self.consts[nodes[use2[0]].input[1]]).reshape(-1)]
                        o_drop.append(use2[0])
                        out = nodes[use2[0]].output[0]
            if not a_chain and not b_chain and out_perm is None:
                continue
            v = pb.Node("_MatMulView", [a_src, b_src], [out])
            v.a_chain, v.b_chain, v.out_perm, v.out_reshape = a_chain, b_chain, out_perm, out_reshape
            replace[idx] = [v]
            drop.update(a_drop + b_drop + o_drop)
        return [m for idx, n in enumerate(nodes) for m in (replace.get(idx, [n]) if idx not in drop or idx in replace else [])]

    def fold_linear_residuals(self):
        """fused_quantized_linear whose result is read once, by an `add` / `add3` with tensors that already exist: the Adds move
        into the GEMM's store (`fused_quantized_linear_residual`, ((lin + r1) + r2) in the Adds' own order)."""
        sts = self.statements
        outs = {sanitize(o) for o in self.outputs}
        defined_at, readers = {}, {}

        def refs(n, acc):
            if isinstance(n, dict):
                for key in ("ref", "ints"):
                    if isinstance(n.get(key), str):
                        acc.append(n[key])
                for v in n.values():
                    refs(v, acc)
            elif isinstance(n, list):
                for v in n:
                    refs(v, acc)
            return acc
        for i, st in enumerate(sts):
            for o in st["out"]:
                defined_at[o] = i
            for r in refs(st.get("args", st.get("in")), []):
                readers.setdefault(r, []).append(i)
        dead = set()
        for i, st in enumerate(sts):
            if st.get("fn") != "fused_quantized_linear":
                continue
            lin = st["out"][0]
            rd = readers.get(lin, [])
            if len(rd) != 1 or lin in outs or rd[0] in dead:
                continue
            use = sts[rd[0]]
            ops = use.get("args", [])
            if use.get("fn") not in ("add", "add3") or not all(isinstance(a, dict) and "ref" in a for a in ops):
                continue
            names = [a["ref"] for a in ops]
            if names.count(lin) != 1 or (use["fn"] == "add3" and names.index(lin) == 2):
                continue                      # (r1 + r2) + lin is a different rounding order
            res = [nm for nm in names if nm != lin]
            if any(defined_at.get(nm, -1) >= i for nm in res):   # graph inputs are defined "before everything" (-1)
                continue
            st["fn"] = "fused_quantized_linear_residual"
            st["args"] = st["args"] + [{"ref": res[0]}, ({"ref": res[1]} if len(res) > 1 else {"none": 1})]
            st["out"] = list(use["out"])
            dead.add(rd[0])
        self.statements[:] = [st for i, st in enumerate(sts) if i not in dead]

    def fold_ffn(self):
        """fused_quantized_linear(x, W1.., relu) whose result is read once, by the input of a second
        fused_quantized_linear[_residual]: ONE `fused_ffn_quantized` statement (lele_hip_fused_ffn_quantized: the f32 hidden
        tensor need not exist).  The run-time form issues the two calls itself when the shapes do not suit the fused route, so
        the rewrite is always legal.  The merged statement stands where the second linear stood (its residuals exist there)."""
        sts = self.statements
        outs = {sanitize(o) for o in self.outputs}
        readers = {}

        def refs(n, acc):
            if isinstance(n, dict):
                for key in ("ref", "ints"):
                    if isinstance(n.get(key), str):
                        acc.append(n[key])
                for v in n.values():
                    refs(v, acc)
            elif isinstance(n, list):
                for v in n:
                    refs(v, acc)
            return acc
        for i, st in enumerate(sts):
            for r in refs(st.get("args", st.get("in")), []):
                readers.setdefault(r, []).append(i)
        dead = set()
        for i, st in enumerate(sts):
            if st.get("fn") != "fused_quantized_linear" or st["args"][5] != {"bool": True}:
                continue
            hid = st["out"][0]
            rd = readers.get(hid, [])
            if len(rd) != 1 or hid in outs or rd[0] <= i:
                continue
            second = sts[rd[0]]
            if second.get("fn") not in ("fused_quantized_linear", "fused_quantized_linear_residual"):
                continue
            b = second["args"]
            if b[0] != {"ref": hid} or refs(b[1:], []).count(hid):
                continue
            a = st["args"]
            res = b[6:8] if second["fn"].endswith("_residual") else [{"none": 1}, {"none": 1}]
            second["fn"] = "fused_ffn_quantized"
            second["args"] = [a[0], a[1], a[2], a[3], a[4], b[1], b[2], b[3], b[4], b[5]] + list(res)
            dead.add(i)
        self.statements[:] = [st for i, st in enumerate(sts) if i not in dead]

    def fold_ln_epilogue(self):
        """A projection's sum and the LayerNorm that reads it, as one statement with two results:

            x1 = fused_quantized_linear_residual(..., res1, res2);  x1n = layer_norm(x1, g, b, -1, eps)
              -> [x1, x1n] = fused_quantized_linear_residual_ln(..., res1, res2, g, b, eps)
            y = fused_ffn_quantized(...);  yn = layer_norm(y, g, b, -1, eps)   ->   [y, yn] = fused_ffn_quantized_ln(..., g, b, eps)

        and, where res1 is a `depthwise_conv1d_tlc(.., relu = false, add_input = true)` that nothing else reads (the FSMN memory block
        of a SAN-M layer), the three of them as `sanm_out_block` (lele_hip_sanm_out_block).  x1 may have other readers (it is the residual
        stream); the LayerNorm must normalise over the last axis with constant scale and bias.  The run-time forms issue the
        separate calls themselves wherever the one-launch kernel does not take the shapes, so the rewrite is always legal; both
        are bit-identical to the sequence (tests/test_quant.py)."""
        sts = self.statements
        outs = {sanitize(o) for o in self.outputs}
        readers = {}

        def refs(n, acc):
            if isinstance(n, dict):
                for key in ("ref", "ints"):
                    if isinstance(n.get(key), str):
                        acc.append(n[key])
                for v in n.values():
                    refs(v, acc)
            elif isinstance(n, list):
                for v in n:
                    refs(v, acc)
            return acc
        producer = {}
        for i, st in enumerate(sts):
            for o in st.get("out", []):
                producer[o] = i
            for r in refs(st.get("args", st.get("in")), []):
                readers.setdefault(r, []).append(i)
        dead = set()
        for i, st in enumerate(sts):
            if st.get("fn") not in ("fused_quantized_linear_residual", "fused_ffn_quantized") or i in dead:
                continue
            x1 = st["out"][0]
            lns = [j for j in readers.get(x1, []) if j > i and sts[j].get("fn") == "layer_norm" and sts[j]["args"][0] == {"ref": x1}
                   and sts[j]["args"][3] == {"int": -1} and all("ref" not in a for a in sts[j]["args"][1:3])]
            if len(lns) != 1:
                continue
            ln = sts[lns[0]]
            g, b, eps = ln["args"][1], ln["args"][2], ln["args"][4]
            if st["fn"] == "fused_ffn_quantized":      # the feed-forward block + the NEXT half-layer's LayerNorm
                st["fn"] = "fused_ffn_quantized_ln"
                st["args"] = st["args"][:12] + [g, b, eps]
                st["out"] = [x1, ln["out"][0]]
                st["bufs"] = 2
                dead.add(lns[0])
                continue
            a = st["args"]          # input, w, scale, zero, bias, relu, res1, res2
            mem = a[6].get("ref") if isinstance(a[6], dict) else None
            tlc = sts[producer[mem]] if mem in producer else None
            if (tlc is not None and tlc.get("fn") == "depthwise_conv1d_tlc" and readers.get(mem, []) == [i] and mem not in outs
                    and tlc["args"][5] == {"bool": False} and tlc["args"][7] == {"bool": True} and producer[mem] not in dead):
                t = tlc["args"]     # x, w, bias, pl, pr, relu, x_offset, add_input
                st["fn"] = "sanm_out_block"
                st["args"] = a[:6] + [t[0], t[1], t[2], t[6], t[3], t[4], a[7], g, b, eps]
                dead.add(producer[mem])
            else:
                st["fn"] = "fused_quantized_linear_residual_ln"
                st["args"] = a[:8] + [g, b, eps]
            st["out"] = [x1, ln["out"][0]]
            st["bufs"] = 2
            dead.add(lns[0])
        self.statements[:] = [st for i, st in enumerate(sts) if i not in dead]

    def fold_attention(self):
        """matmul_view(Q view, K^T view) -> softmax_scaled -> matmul_view(P, V view [, out_perm, out_reshape]) with private
        intermediates becomes ONE `attention_view` statement (lele_hip_attention_view: the score / probability tensors stay on
        chip).  Shapes are not known here: the run-time form checks the geometry and issues the three calls itself when the
        kernel does not take it -- including chains that are not one strided view of their source, which the three calls
        materialise exactly as the unfused statements would (kernels.attention_view, plan_runner.hpp attention_view)."""
        sts = self.statements
        outs = {sanitize(o) for o in self.outputs}
        readers = {}

        def refs(n, acc):
            if isinstance(n, dict):
                for key in ("ref", "ints"):
                    if isinstance(n.get(key), str):
                        acc.append(n[key])
                for v in n.values():
                    refs(v, acc)
            elif isinstance(n, list):
                for v in n:
                    refs(v, acc)
            return acc
        for i, st in enumerate(sts):
            for r in refs(st.get("args", st.get("in")), []):
                readers.setdefault(r, []).append(i)
        dead = set()
        for i, st in enumerate(sts):
            if st.get("fn") != "matmul_view" or i in dead:
                continue
            a = st["args"]
            if "none" not in a[4] or "none" not in a[5]:
                continue
            sc = st["out"][0]
            rd = readers.get(sc, [])
            if len(rd) != 1 or sc in outs or sts[rd[0]].get("fn") != "softmax_scaled":
                continue
            sm = sts[rd[0]]
            if sm["args"][0] != {"ref": sc} or sm["args"][2] != {"int": -1}:
                continue
            pr = sm["out"][0]
            rd2 = readers.get(pr, [])
            if len(rd2) != 1 or pr in outs or sts[rd2[0]].get("fn") != "matmul_view":
                continue
            pv = sts[rd2[0]]
            b = pv["args"]
            if b[0] != {"ref": pr} or b[1] != {"chain": []} or not (rd[0] < rd2[0]):
                continue
            # the V operand must already exist where the first product stands (the fused statement takes its place)
            st["fn"] = "attention_view"
            st["args"] = [a[0], a[1], a[2], a[3], b[2], b[3], sm["args"][1], b[4], b[5]]
            st["out"] = list(pv["out"])
            # everything the fused statement reads beyond the first product's own operands is hoisted to the first product's
            # position: V and its view chain, the softmax scale, out_perm / out_reshape (run-time `ints` refs included)
            moved_ok = True
            for r in refs([b[2], b[3], b[4], b[5], sm["args"][1]], []):
                if any(r in s2["out"] for s2 in sts[i + 1:rd2[0]]):
                    moved_ok = False
            if not moved_ok:      # one of them is produced between the two products: keep the sequence
                st["fn"], st["args"], st["out"] = "matmul_view", a, [sc]
                continue
            dead.update((rd[0], rd2[0]))
        self.statements[:] = [st for i, st in enumerate(sts) if i not in dead]

    # ---------------------------------------------------------------------------------------- driver
    def lower_graph(self, graph_nodes):
        nodes = self.fold(graph_nodes)
        cnt = self.uses(nodes)
        if self.extra_fusions:
            nodes = self.push_views(nodes, cnt)
            cnt = self.uses(nodes)
        k = 0
        while k < len(nodes):
            n = nodes[k]
            if n.op_type == "If":
                self.lower_if(n)
                k += 1
                continue
            ins = [i for i in n.input if i]
            # integer / tiny-float side arithmetic stays on the host; a float table (an embedding) is device data
            host_ready = all((i in self.consts and (self.consts[i].dtype.kind in "iub" or self.consts[i].size <= 16))
                             or self.place.get(i) == "host" for i in ins)
            if n.op_type == "ConstantOfShape":  # a float fill is tensor data (e.g. an RNN's initial state): a device fill
                v = _attrs(n).get("value")
                host_ready = host_ready and v is not None and np.asarray(v).dtype.kind in "iub"
            if n.op_type in ("Shape", "Size") or (host_ready and n.op_type in HOST_OPS):
                self.emit_host(n, _attrs(n))
                k += 1
                continue
            m = self.match(nodes, k, cnt)
            if m:
                used, emitter = m
                emitter()
                k += used
                continue
            self.lower_node(n)
            k += 1
        if self.extra_fusions:
            self.fold_linear_residuals()
            self.fold_ffn()
            if getattr(self, "half_layer_folds", True):   # False: round 5's eight statements a layer (the FSMN block a branch of its own: tests/test_lanes.py)
                self.fold_ln_epilogue()     # after fold_ffn: the feed-forward block's second linear belongs to that fold
            self.fold_attention()

    def lower_if(self, node):
        """ops/control_flow.rs:18-150: `let (outs) = if cond.data[0] != 0 { then } else { else }` -- the condition is read on
        the host when the statement runs, the branches are statement lists of their own (own workspace slots), and the
        results are owned copies (`.to_owned()`): the statement has one output buffer per device result."""
        at = _attrs(node)
        self.if_count[0] += 1
        tag = "if%d" % self.if_count[0]
        arms, kinds = {}, None
        for key in ("then", "else"):
            g = at[key + "_branch"]
            if len(g.output) != len(node.output):
                raise CompileError("If %r: the %s branch has %d outputs, the node %d" % (node.name, key, len(g.output), len(node.output)))
            ch = Lowering(self.model, self.name, self.extra_fusions, graph=g, parent=self)
            ch.half_layer_folds = getattr(self, "half_layer_folds", True)
            ch.lower_graph(list(g.node))
            results, k = [], []
            for o in g.output:
                if o.name in ch.consts:
                    arr = np.asarray(ch.consts[o.name])
                    if arr.dtype.kind in "iub":
                        results.append({"const": arr.tolist(), "dtype": "i64"})
                        k.append("host")
                    else:
                        results.append(ch.weight(o.name, True))
                        k.append("dev")
                else:
                    results.append({"ref": sanitize(o.name)})
                    k.append(ch.place.get(o.name, "dev"))
            if kinds is not None and k != kinds:
                raise CompileError("If %r: the branches disagree on which results are host values (%s vs %s)" % (node.name, kinds, k))
            kinds = k
            inner = allocate(ch.statements, [r["ref"] for r in results if "ref" in r])
            rename = {s_: "%s%s_%s" % (tag, key[0], s_) for s_ in inner}
            for st in ch.statements:
                if "slots" in st:
                    st["slots"] = [rename[s_] for s_ in st["slots"]]
            self.branch_slots += [rename[s_] for s_ in inner]
            arms[key] = {"statements": ch.statements, "results": results}
        outs = [sanitize(o) for o in node.output]
        dev_out = [o for o, k in zip(outs, kinds) if k == "dev"]
        reads = [{"ref": sanitize(i)} for i in node.input if i and i not in self.consts]
        self.statements.append({"op": "if", "out": outs, "dev_out": dev_out, "kinds": kinds, "bufs": len(dev_out),
                                "cond": {"ref": sanitize(node.input[0])}, "args": reads, "then": arms["then"], "else": arms["else"]})
        for o, k in zip(node.output, kinds):
            self.place[o] = k

    def run(self):
        self.lower_graph(list(self.model.graph.node))
        for o in self.outputs:
            if o in self.consts:
                raise CompileError("graph output %s is a constant" % o)
        slots = allocate(self.statements, [sanitize(o) for o in self.outputs])
        plan = {"source": self.name, "format": "lele_amd.plan/2", "inputs": [sanitize(v.name) for v in self.inputs],
                "input_info": [{"name": sanitize(v.name), "dtype": "i64" if self.place[v.name] == "host" else "f32", "shape": v.shape}
                               for v in self.inputs],
                "outputs": [sanitize(o) for o in self.outputs], "slots": slots + self.branch_slots, "statements": self.statements,
                "weights": {}}
        seen = {}

        def walk(n):
            if isinstance(n, dict):
                if "weight" in n:
                    kind, off, ln, shape = n["weight"]
                    seen["%d:%s:%s" % (off, kind, "x".join(map(str, shape)))] = [kind, off, ln, shape]
                for v in n.values():
                    walk(v)
            elif isinstance(n, list):
                for v in n:
                    walk(v)
        walk(self.statements)
        plan["weights"] = seen
        return plan, bytes(self.packer.blob)


HOST_OPS = {"Identity", "Shape", "Size", "Unsqueeze", "Squeeze", "Concat", "Gather", "Cast", "Add", "Sub", "Mul", "Div", "Equal", "Less",
            "Greater", "Max", "Min", "Neg", "Reshape", "Slice", "Range", "ConstantOfShape", "Expand", "Where", "Transpose", "Tile", "Not"}


def allocate(statements, outputs):
    """Workspace slots by liveness (src/compiler/mod.rs:148-290): a value's slot returns to the free heap after its last
    reader; views extend the life of the buffer they share; a statement never writes a slot it (or one of the five
    statements before it, which may be fused with it upstream) reads.  Adds "slots" to every device statement in place and
    returns the slot names."""
    last_use, owner, maybe = {}, {}, {}
    INF = len(statements) + 1

    def refs(n, acc):
        if isinstance(n, dict):
            for key in ("ref", "ints"):
                if key in n and isinstance(n[key], str):
                    acc.append(n[key])
            if isinstance(n.get("refs"), list):   # lifted plans: a concat's operand list
                acc += n["refs"]
            for v in n.values():
                refs(v, acc)
        elif isinstance(n, list):
            for v in n:
                refs(v, acc)
        return acc

    reads = []
    for i, st in enumerate(statements):
        r = refs(st.get("args", st.get("in")), [])
        # channel views (plan.fold_channel_views): a `chview` shares its source's buffer, a statement with a `window` writes into
        # (and its result shares) the buffer of the enclosing tensor
        view_src = st.get("src") if st["op"] == "chview" else (st["window"]["of"] if "window" in st else None)
        if view_src is not None:
            r = r + [view_src] + list(st.get("after", []))
        reads.append(r)
        for name in r:
            last_use[name] = i
        if view_src is not None:
            for o in st["out"]:
                owner[o] = owner.get(view_src, view_src)
        elif st["op"] == "call" and st.get("bufs", 1) == 0:  # view: shares its first tensor operand's buffer
            src = r[0] if r else None
            owner[st["out"][0]] = owner.get(src, src)
        elif st.get("may_alias") and r:                      # may turn out to be a view of its first operand at run time
            maybe[st["out"][0]] = r[0]
    for o in outputs:
        last_use[o] = INF
    # a buffer lives as long as its longest-lived view
    for i in range(len(statements) - 1, -1, -1):
        for o in statements[i]["out"]:
            root = owner.get(o)
            if root is None and o in maybe:
                root = owner.get(maybe[o], maybe[o])
            if root is not None:
                last_use[root] = max(last_use.get(root, -1), last_use.get(o, -1))
    free, active, slot_of, n_slots = [], {}, {}, 0
    import heapq
    for i, st in enumerate(statements):
        for name in [n for n, _s in active.items() if last_use.get(n, -1) < i]:
            heapq.heappush(free, active.pop(name))
        if st["op"] not in ("call", "if", "reserve") or st.get("bufs", 1) == 0:
            continue
        busy = {slot_of[owner.get(r, r)] for j in range(max(0, i - 5), i + 1) for r in reads[j] if owner.get(r, r) in slot_of}
        st["slots"] = []
        for b in range(st["bufs"]):
            deferred, pick = [], None
            while free:
                s = heapq.heappop(free)
                if s in busy:
                    deferred.append(s)
                else:
                    pick = s
                    break
            for s in deferred:
                heapq.heappush(free, s)
            if pick is None:
                pick, n_slots = n_slots, n_slots + 1
            st["slots"].append("buf_%d" % pick)
            busy.add(pick)
            names = st.get("dev_out", st["out"])   # an `if` owns one buffer per DEVICE result
            name = names[b] if b < len(names) else None
            if name is not None:
                slot_of[name] = pick
                active[name] = pick
    return ["buf_%d" % s for s in range(n_slots)]


def compile_model(model, name="model", extra_fusions=True, bind=None, half_layer_folds=True):
    """ONNX model (bytes, path or onnx_pb.Model) -> (plan dict, weights.bin bytes).  extra_fusions=False keeps to the
    patterns lele's own compiler has (patterns.rs); the extra fused forms are bit-identical to what they replace.
    bind={input: value} fixes graph inputs at compile time (an `If` on them then inlines the taken branch)."""
    if not isinstance(model, pb.Model):
        model = pb.load(model)
    low = Lowering(model, name, extra_fusions, bind=bind)
    low.half_layer_folds = bool(half_layer_folds)
    return low.run()
