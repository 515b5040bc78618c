"""Plan execution: the statement list produced by `lele_amd.compiler` (from ONNX) or by tools/lift_generated.py (from a
lele-generated Rust model) run through the operator mirror, one C-ABI call per statement, recordable as a hipGraph.

A plan is JSON: `inputs`, `outputs`, `slots` (workspace buffers, lele's `ws.buf_k`), `weights` (views into a lele
`<model>_weights.bin`: kind, byte offset, byte length, shape -- src/compiler/mod.rs:1135-1233) and `statements`:

  {"op": "call", "out": [...], "fn": "<lele::kernels name>", "args": [...], "slots": ["buf_3"]}   a kernel call
  {"op": "host", "out": [...], "onnx": "<op>", "in": [...], "attrs": {...}}                        integer side arithmetic (hostops)
  {"op": "ints" | "newbuf" | "swap_remove" | "alias", ...}                                       forms lele's emitters produce

Argument nodes: {"ref"}, {"weight"}, {"list"}, {"int"|"float"|"bool"|"str"}, {"none"}, {"some"}, {"slot"}, {"buf"},
{"ints": name} (a host integer value as a list), {"first": node} (first element of an integer list),
{"array": [...], "dtype"} (a literal host tensor), {"chain": [...]} (the view steps of a `view_copy`).
"""
import time

import numpy as np

from .compiler import hostops

WEIGHT_DTYPES = {"weight_f32": "<f4", "weight_i64": "<i8", "weight_i64_f32": "<i8", "weight_i32": "<i4", "weight_i32_i64": "<i4",
                 "weight_i32_f32": "<i4", "weight_u8": "u1", "weight_i8": "i1", "weight_f16": "<f2", "weight_f64": "<f8"}


def load_weights_bin(plan, data):
    """decode the views of a lele `<model>_weights.bin` (path or bytes).  As lele's accessors do (src/compiler/mod.rs:1135-1233),
    u8 / i8 / f16 / f64 / i32 tensors are handed to the kernels as f32 values, i64 stays i64 unless the view says `_f32`."""
    if not isinstance(data, (bytes, bytearray, memoryview)):
        data = open(data, "rb").read()
    out = {}
    for key, view in plan["weights"].items():
        kind, off, ln, shape = view if len(view) == 4 else (view[0], int(key), view[1], view[2])  # lifted plans key by offset
        if kind not in WEIGHT_DTYPES:
            raise ValueError("weights view kind %r is not handled" % kind)
        a = np.frombuffer(data[off:off + ln], WEIGHT_DTYPES[kind])
        if kind in ("weight_i64", "weight_i32_i64"):
            a = a.astype(np.int64)
        elif kind == "weight_i32":
            a = a.astype(np.int64)
        else:
            a = a.astype(np.float32)
        out[key if len(view) == 4 else int(key)] = np.array(a).reshape(shape if shape else ())
    return out


def weight_key(node):
    kind, off, _ln, shape = node
    return "%d:%s:%s" % (off, kind, "x".join(map(str, shape)))


def fuse_sigmoid_mul(plan, shapes):
    """Peephole for plans lifted from lele-generated Rust: `s = sigmoid(x)` ... `z = mul(x, s)` (s read nowhere else) becomes
    `z = silu(x)` in the mul's place and buffer -- lele's own "SiLU" pattern, which its window-based matcher misses when the
    branches of a block are interleaved (the generated Yolo26n-seg has 39 such pairs).  Only where the element count is a
    multiple of 8 (`shapes`: value name -> shape, recorded by a first run): there x * sigmoid(x) and silu(x) are the same bits;
    on a ragged tail the reference's silu divides instead of multiplying by the reciprocal.  Returns a new plan."""
    sts = plan["statements"]

    def refs(n, acc):
        if isinstance(n, dict):
            if isinstance(n.get("ref"), str):
                acc.append(n["ref"])
            acc += n.get("refs", [])
            for v in n.values():
                refs(v, acc)
        elif isinstance(n, list):
            for v in n:
                refs(v, acc)
        return acc
    readers = {}
    for i, st in enumerate(sts):
        for r in refs(st.get("args"), []):
            readers.setdefault(r, []).append(i)
    drop, replace = set(), {}
    for i, st in enumerate(sts):
        if st.get("fn") != "sigmoid" or "ref" not in st["args"][0]:
            continue
        x, sg = st["args"][0]["ref"], st["out"][0]
        rd = readers.get(sg, [])
        if len(rd) != 1 or sg in plan["outputs"] or int(np.prod(shapes.get(x, [1]))) % 8:
            continue
        mul = sts[rd[0]]
        ops = [a.get("ref") for a in mul.get("args", []) if isinstance(a, dict) and "ref" in a]
        if mul.get("fn") != "mul" or sorted(ops) != sorted([x, sg]):
            continue
        outbuf = [a for a in mul["args"] if "slot" in a or "buf" in a]
        replace[rd[0]] = dict(mul, fn="silu", args=[{"ref": x}] + outbuf)
        drop.add(i)
    new = dict(plan)
    new["statements"] = [replace.get(i, st) for i, st in enumerate(sts) if i not in drop]
    return new


def replan_lifted(plan, shapes):
    """A plan lifted from lele-generated Rust, re-planned: buffers re-assigned by this library's liveness allocator
    (lele_amd.compiler.lower.allocate) instead of lele's, which makes statement-level fusions safe that move a result into
    another statement -- here `conv2d` followed by a private `silu` (itself from fuse_sigmoid_mul) -> `conv2d_silu`, wherever
    the plane size is a multiple of 8 (then the convolution's SiLU epilogue and the separate kernel are the same bits).
    `split_owned` + `swap_remove` become one `split` with named outputs.  Returns a format-2 plan (runs in both runners)."""
    from .compiler.lower import allocate
    src = plan["statements"]
    ints = {}                                          # the generated code re-binds the same names: track the latest value
    out, i = [], 0
    while i < len(src):
        st = src[i]
        op = st["op"]
        if op == "newbuf" or op == "swap_remove":
            i += 1
            continue
        if op == "alias":
            out.append({"op": "call", "out": list(st["out"]), "fn": "identity", "args": [{"ref": st["src"]}], "bufs": 0})
        elif op == "ints":
            ints[st["out"][0]] = st["value"]
            out.append(dict(st))
        elif st.get("fn") == "split_owned":
            lst = st["out"][0]
            sizes = ints[st["args"][2]["ref"]]
            order = list(range(len(sizes)))           # Vec::swap_remove semantics, simulated on the index list
            names = {}
            j = i + 1
            while j < len(src) and order:
                nx = src[j]
                if nx["op"] == "swap_remove" and nx["list"] == lst:
                    k = nx["index"]
                    names[order[k]] = nx["out"][0]
                    order[k] = order[-1]
                    order.pop()
                elif nx.get("fn") == "split_owned" and nx["out"][0] == lst:
                    break
                j += 1
            outs = [names.get(k, "%s__unused%d_%d" % (lst, i, k)) for k in range(len(sizes))]
            out.append({"op": "call", "out": outs, "fn": "split", "args": [st["args"][0], st["args"][1], {"list": [{"int": int(v)} for v in sizes]}],
                        "bufs": len(sizes)})
        else:
            args = [a for a in st["args"] if not (isinstance(a, dict) and ("slot" in a or "buf" in a))]
            nb = len(st["args"]) - len(args)
            if st["fn"] in ("reshape", "flatten", "unsqueeze", "squeeze", "identity"):
                nb = 0
            out.append({"op": "call", "out": list(st["out"]), "fn": st["fn"], "args": args, "bufs": nb})
        i += 1
    # conv2d -> private silu  =>  conv2d_silu
    readers = {}

    def refs(n, acc):
        if isinstance(n, dict):
            if isinstance(n.get("ref"), str):
                acc.append(n["ref"])
            acc += n.get("refs", [])
            for v in n.values():
                refs(v, acc)
        elif isinstance(n, list):
            for v in n:
                refs(v, acc)
        return acc
    for k, st in enumerate(out):
        for r in refs(st.get("args"), []):
            readers.setdefault(r, []).append(k)
    dead = set()
    for k, st in enumerate(out):
        if st.get("fn") != "conv2d":
            continue
        y = st["out"][0]
        rd = readers.get(y, [])
        shp = shapes.get(y, [])
        if len(rd) != 1 or y in plan["outputs"] or len(shp) != 4 or (shp[2] * shp[3]) % 8 or out[rd[0]].get("fn") != "silu":
            continue
        st["fn"], st["out"] = "conv2d_silu", list(out[rd[0]]["out"])
        dead.add(rd[0])
    out = [st for k, st in enumerate(out) if k not in dead]
    slots = allocate(out, list(plan["outputs"]))
    weights = {}
    for off, (kind, ln, shape) in plan["weights"].items():
        weights[weight_key([kind, int(off), ln, shape])] = [kind, int(off), ln, shape]
    return {"source": plan["source"], "format": "lele_amd.plan/2", "inputs": list(plan["inputs"]), "outputs": list(plan["outputs"]),
            "slots": slots, "statements": out, "weights": weights}


class Runner:
    def __init__(self, plan, weights, ctx):
        from . import kernels as K
        from ._lib import Weight
        self.plan, self.K, self.ctx = plan, K, ctx
        self.v2 = plan.get("format") == "lele_amd.plan/2"   # compiled plans key weights by (offset, kind, shape); lifted ones by offset
        self.raw = {(k if self.v2 else int(k)): v for k, v in weights.items()}
        self.W = {k: (Weight(a) if a.dtype != np.int64 else a) for k, a in self.raw.items()}
        self.ws = {s: ctx.buf() for s in plan["slots"]}
        self.extra = {}
        self.calls = 0
        self.profile = None
        self.stmt_index = 0
        self.shapes = None  # set to {} to record the shape of every tensor value of the next run
        self.taps = None    # set to {name: None, ...}: host copies of those results are left there by the next run (tests)

    def _wkey(self, node):
        return weight_key(node) if self.v2 else node[1]

    @staticmethod
    def _host_ints(v):
        from .tensor import TensorView
        if isinstance(v, TensorView):
            v = v.numpy()
        return [int(x) for x in np.asarray(v).reshape(-1)]

    def val(self, n, env):
        if "ref" in n:
            return env[n["ref"]]
        if "refs" in n:
            return [env[r] for r in n["refs"]]
        if "weight" in n:
            return self.W[self._wkey(n["weight"])]
        if "weight_scalar" in n:
            return int(np.asarray(self.raw[self._wkey(n["weight_scalar"])]).reshape(-1)[0])
        if "weight_list" in n:
            a = np.asarray(self.raw[self._wkey(n["weight_list"])]).reshape(-1)
            return [float(v) for v in a] if a.dtype == np.float32 else [int(v) for v in a]
        if "ints" in n:
            return self._host_ints(env[n["ints"]])
        if "chain" in n:
            return n["chain"]
        if "array" in n:
            return np.asarray(n["array"], np.int64 if n.get("dtype") == "i64" else np.float32)
        if "first" in n:
            return self.val(n["first"], env)[0]
        if "some" in n:
            return self.val(n["some"], env)
        if "none" in n:
            return None
        if "list" in n:
            return [self.val(v, env) for v in n["list"]]
        for k in ("int", "float", "bool", "str"):
            if k in n:
                return n[k]
        raise ValueError(n)

    @staticmethod
    def moves_only_unit_axes(shape, perm):
        """a transpose that changes no byte: the axes longer than 1 keep their order"""
        perm = [p + len(shape) if p < 0 else p for p in perm]
        kept = [p for p in perm if shape[p] != 1]
        return len(perm) == len(shape) and kept == sorted(kept)

    def call(self, f, fn, pos, bufs, key=None, may_alias=False):
        ctx = self.ctx
        if fn == "transpose" and may_alias and self.moves_only_unit_axes(list(pos[0].shape), pos[1]):
            self.calls -= 1                                  # a view, not a kernel (the plan reserved for both: lower.py, allocate)
            perm = [p + len(pos[0].shape) if p < 0 else p for p in pos[1]]
            return self.K.reshape(pos[0], [int(pos[0].shape[p]) for p in perm])
        if fn == "split_owned":  # owned results: one persistent device buffer per output of this statement
            outs = self.extra.setdefault(("split", key), [ctx.buf() for _ in pos[2]])
            return list(f(pos[0], pos[1], pos[2], outputs=outs, ctx=ctx))
        if fn == "split":
            return list(f(pos[0], pos[1], pos[2], outputs=bufs, ctx=ctx))
        if fn == "topk":
            return f(pos[0], pos[1], pos[2], pos[3], pos[4], out_values=bufs[0], out_indices=bufs[1], ctx=ctx)
        if fn in ("lstm", "gru", "dynamic_quantize_linear"):
            return f(*pos, outs=bufs, ctx=ctx)
        if fn in ("reshape", "flatten", "unsqueeze", "squeeze", "identity"):
            return f(*pos)
        return f(*pos, out=bufs[0], ctx=ctx) if bufs else f(*pos, ctx=ctx)

    def host(self, st, env):
        from .tensor import TensorView
        ins = []
        for n in st["in"]:
            if n is None:
                ins.append(None)
            elif "const" in n:
                ins.append(np.asarray(n["const"], np.int64 if n["dtype"] == "i64" else np.float32))
            else:
                v = env[n["ref"]]
                if st["onnx"] in ("Shape", "Size"):  # only the shape is needed: never a device read
                    v = np.empty([int(d) for d in v.shape], np.uint8) if isinstance(v, TensorView) else np.asarray(v)
                elif isinstance(v, TensorView):
                    v = v.numpy()
                ins.append(np.asarray(v))
        res = hostops.evaluate(st["onnx"], ins, st["attrs"])
        if res is None:
            raise RuntimeError("host statement %s: no host evaluator" % st["onnx"])
        for name, r in zip(st["out"], res):
            env[name] = np.asarray(r)

    def run(self, inputs):
        env = dict(inputs)
        self.stmt_index = 0
        self.exec(self.plan["statements"], env)
        if self.shapes is not None:
            for name, v in env.items():
                if hasattr(v, "shape"):
                    self.shapes[name] = [int(d) for d in v.shape]
        return [env[o] for o in self.plan["outputs"]]

    def if_(self, st, env):
        """`let (outs) = if cond.data[0] != 0 {..} else {..}` (ops/control_flow.rs:18-150): the condition is read on the host --
        a device value is fetched, which waits for the stream (and cannot happen inside a graph capture, where the library
        refuses it) -- then the taken branch's statements run and its device results are copied into this statement's
        buffers, as the generated code's `.to_owned()` does."""
        from .tensor import TensorView
        c = self.val(st["cond"], env)
        c = c.numpy() if isinstance(c, TensorView) else np.asarray(c)
        arm = st["then"] if c.size and c.reshape(-1)[0] != 0 else st["else"]
        self.exec(arm["statements"], env)
        bufs = [self.ws[s] for s in st.get("slots", [])]
        k = 0
        for name, res, kind in zip(st["out"], arm["results"], st["kinds"]):
            if kind == "host":
                env[name] = np.asarray(res["const"], np.int64 if res.get("dtype") == "i64" else np.float32) if "const" in res else env[res["ref"]]
            else:
                self.calls += 1
                env[name] = self.K.view_copy(self.val(res, env), [], out=bufs[k], ctx=self.ctx)
                k += 1

    def exec(self, statements, env):
        K, ctx = self.K, self.ctx
        for st in statements:
            self.stmt_index += 1
            op = st["op"]
            if op == "if":
                self.if_(st, env)
            elif op == "ints":
                env[st["out"][0]] = st["value"]
            elif op == "newbuf":
                self.extra.setdefault(st["out"][0], ctx.buf())
                env[st["out"][0]] = self.extra[st["out"][0]]
            elif op == "swap_remove":  # Vec::swap_remove: take element i, move the last element into its place
                lst = env[st["list"]]
                i = st["index"]
                env[st["out"][0]] = lst[i]
                lst[i] = lst[-1]
                lst.pop()
            elif op == "alias":
                env[st["out"][0]] = env[st["src"]]
            elif op == "host":
                self.host(st, env)
            else:
                fn, args = st["fn"], st["args"]
                pos = []
                bufs = [self.ws[s] for s in st.get("slots", [])]
                for a in args:
                    if "slot" in a:
                        bufs.append(self.ws[a["slot"]])
                    elif "buf" in a:
                        bufs.append(env[a["buf"]])
                    else:
                        pos.append(self.val(a, env))
                f = getattr(K, fn)
                self.calls += 1
                try:
                    t0 = time.perf_counter() if self.profile is not None else 0.0
                    res = self.call(f, fn, pos, bufs, key=st["out"][0] + str(self.stmt_index), may_alias=bool(st.get("may_alias")))
                    if self.profile is not None:
                        ctx.sync()
                        self.profile[fn] = self.profile.get(fn, 0.0) + time.perf_counter() - t0
                except Exception as e:  # noqa: BLE001
                    shp = [getattr(p, "shape", p) if not isinstance(p, list) else [getattr(q, "shape", q) for q in p] for p in pos]
                    raise RuntimeError("statement %s = %s(...) failed with %s; argument shapes/values: %s" % (st["out"], fn, e, shp))
                if len(st["out"]) == 1:
                    env[st["out"][0]] = res
                else:
                    for name, r in zip(st["out"], res):
                        env[name] = r
                if self.taps:  # test aid: host copies of named results, taken before their workspace slot is reused
                    for name in st["out"]:
                        if name in self.taps:
                            self.taps[name] = env[name].numpy().copy()
