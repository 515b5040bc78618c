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
import os
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
    the plane size is a multiple of 8 (then the REPLICA form of the convolution's SiLU epilogue, LELE_HIP_CONV_SILU_EXACT=1, and the
    separate kernel are the same bits; the default epilogue is within 1e-5 of them).
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


def rebatch_lifted(plan, n, shapes=None):
    """A re-planned lifted plan (replan_lifted) whose generated source baked batch 1 into its shape literals, for a batch of `n`.
    lele's emitter folds the ONNX shape arithmetic for the export's batch (examples/yolo26n-seg/src/yolo26seg.rs: `reshape(x,
    &[1, 2, 128, 400])`, ...), and its examples loop over images on the host; the data flow itself is batch-agnostic except where
    the exporter's constant folding used N = 1.  Two rewrites, both checked against the per-image run by whoever calls this
    (tools/yolo_lifted_batch.py, tests/test_lift_generated.py):
      * a `reshape` to a literal shape that starts with 1 gets n there (every such reshape in the generated Yolo26n-seg splits or
        merges trailing axes of an [N, ...] tensor; a reshape whose element count would not match fails loudly at run time);
      * `gather(flatten(E, 2), idx, 0)` with E [N, K, 1] and idx [N, K] -- the exporter's form of "row idx[n, i] of image n", which
        flattens the batch away -- becomes `gather_elements(E, unsqueeze(idx, -1), 1)`: the same values at N = 1, per image for N > 1.
    `shapes` (value name -> shape of the BATCH-1 run, Runner.shapes): when given, a statement is rewritten only where the shapes prove
    the pattern -- the reshape's operand has leading dimension 1 and the literal's element count is the operand's (a constant
    reshaped to [1, C, 1, 1] is left alone: its operand is not batched); E is [1, K, 1] and idx [1, K] for the gather.
    Returns a new format-2 plan (buffers re-assigned)."""
    from .compiler.lower import allocate
    import copy
    sts = copy.deepcopy(plan["statements"])

    def numel(shp):
        return int(np.prod(shp, dtype=np.int64)) if shp is not None else None
    prod = {}
    for i, st in enumerate(sts):
        for o in st.get("out", []):
            prod[o] = i
    out, extra = [], 0
    for st in sts:
        st = dict(st)
        st.pop("slots", None)
        if st.get("op") == "call" and st.get("fn") == "reshape" and isinstance(st["args"][1], dict) and "list" in st["args"][1]:
            dims = st["args"][1]["list"]
            ok = bool(dims) and dims[0] == {"int": 1}
            if ok and shapes is not None:
                src = shapes.get(st["args"][0].get("ref")) if isinstance(st["args"][0], dict) else None
                lit = [d.get("int") if isinstance(d, dict) else None for d in dims]
                if any(not isinstance(v, int) for v in lit):     # a {"ref": ...} or float dimension: pattern not proven, leave it alone
                    src = None
                ok = src is not None and len(src) >= 1 and int(src[0]) == 1 and st["args"][0].get("ref") not in plan["weights"] and \
                    (any(v in (-1, 0) for v in lit) or numel(lit) == numel(src))
            if ok:
                st["args"] = [st["args"][0], {"list": [{"int": int(n)}] + list(dims[1:])}] + list(st["args"][2:])
        elif st.get("op") == "call" and st.get("fn") == "gather" and st["args"][2] == {"int": 0} and "ref" in st["args"][0] and "ref" in st["args"][1]:
            src = sts[prod[st["args"][0]["ref"]]] if st["args"][0]["ref"] in prod else None
            ok = src is not None and src.get("fn") == "flatten" and src["args"][1] == {"int": 2} and "ref" in src["args"][0]
            if ok and shapes is not None:
                e, ix = shapes.get(src["args"][0]["ref"]), shapes.get(st["args"][1]["ref"])
                ok = e is not None and ix is not None and len(e) == 3 and int(e[0]) == 1 and int(e[2]) == 1 and list(ix) == [1, int(e[1])]
            if ok:
                extra += 1
                ix = "%s__ix%d" % (st["out"][0], extra)
                out.append({"op": "call", "out": [ix], "fn": "unsqueeze", "args": [st["args"][1], {"list": [{"int": -1}]}], "bufs": 0})
                st["fn"] = "gather_elements"
                st["args"] = [src["args"][0], {"ref": ix}, {"int": 1}]
        out.append(st)
    slots = allocate(out, list(plan["outputs"]))
    new = dict(plan)
    new.update({"statements": out, "slots": slots, "batch": int(n)})
    return new


# ---------------------------------------------------------------------------------------------------------- channel views
# Concat / Split along C of NCHW tensors as VIEWS of one buffer (include/lele_hip.h, LelePitch).  lele copies (manipulation.rs:108-207,
# 1091-1151); the values are the same, so a plan folded here gives the bits of the plan it came from.  Shapes are needed: the pass
# runs on a plan whose value shapes were recorded by one eager run (Runner.shapes) -- the shapes a hipGraph capture freezes anyway.
_CONVS = ("conv2d", "conv2d_silu", "conv2d_fused", "conv2d_res")
_VIEW_WRITERS = _CONVS + ("add", "sub", "mul", "div", "max_pool2d", "resize_nearest")


def _conv_group(st):
    """the literal group of a convolution statement (conv2d_res has the residual in front of the attributes), or None"""
    return _lit_int(st["args"][5 if st["fn"] == "conv2d_res" else 4])


def fuse_residual_adds(sts, shapes, outputs, multi):
    """`c = conv2d*(x, ...)` ... `y = add(c, r)` (either order) -> `y = conv2d_res(x, ..., r, act)` at the convolution's place, when the
    Add is the convolution's only reader, r has the result's shape (no broadcast) and exists before the convolution.  lele's generated
    code keeps the two calls (every bottleneck's `x + cv2(cv1(x))`); lele_hip_conv2d_res forms the same sum from the same two f32
    values.  Returns (statements, number fused)."""
    prod, readers = {}, {}
    for i, st in enumerate(sts):
        for o in st.get("out", []):
            prod[o] = i
        names = _refs(st.get("args", st.get("in")), [])
        if st["op"] == "if":
            names += _refs([st.get("then"), st.get("else"), st.get("cond")], [])
        for r in names:
            readers.setdefault(r, []).append(i)
    drop, put = set(), {}
    for i, st in enumerate(sts):
        if st["op"] != "call" or st.get("fn") != "add" or len(st["out"]) != 1 or st["out"][0] in multi:
            continue
        ab = [a.get("ref") if isinstance(a, dict) else None for a in st["args"][:2]]
        if None in ab or ab[0] == ab[1]:
            continue
        for c, r in (ab, ab[::-1]):
            pi = prod.get(c)
            if pi is None or pi in put or c in multi or r in multi or c in outputs or readers.get(c) != [i]:
                continue
            pst = sts[pi]
            if pst["op"] != "call" or pst.get("fn") not in ("conv2d", "conv2d_silu", "conv2d_fused") or len(pst["out"]) != 1 or pi >= i:
                continue
            sc, sr = shapes.get(c), shapes.get(r)
            if sc is None or len(sc) != 4 or list(sc) != list(sr or []):
                continue
            pr = prod.get(r)
            if pr is not None and pr >= pi:
                continue
            args = [a for a in pst["args"] if not (isinstance(a, dict) and ("slot" in a or "buf" in a))]
            if len(args) < 7 or not isinstance(args[0], dict) or "ref" not in args[0]:
                continue
            if pst["fn"] == "conv2d_fused":
                relu = args[7].get("bool") if len(args) > 7 and isinstance(args[7], dict) else None
                if relu is None:
                    continue
                act = 1 if relu else 0
            else:
                act = 2 if pst["fn"] == "conv2d_silu" else 0
            new = {k: v for k, v in pst.items() if k not in ("slots", "args", "fn", "out")}
            new.update({"fn": "conv2d_res", "out": list(st["out"]), "args": args[:3] + [{"ref": r}] + args[3:7] + [{"int": act}]})
            if "slots" in st:
                new["slots"] = st["slots"]
            put[pi] = new
            drop.add(i)
            break
    out = [put.get(i, st) for i, st in enumerate(sts) if i not in drop]
    return out, len(drop)
_F32_FNS = set(_VIEW_WRITERS) | {"conv_transpose", "silu", "sigmoid", "relu", "tanh", "exp", "sqrt", "softmax", "softmax_scaled", "layer_norm",
                                 "batch_norm", "matmul", "matmul_fused_add", "gemm", "add3", "depthwise_conv1d_tlc"}


def _refs(n, acc):
    if isinstance(n, dict):
        for key in ("ref", "ints"):
            if isinstance(n.get(key), str):
                acc.append(n[key])
        if isinstance(n.get("refs"), list):
            acc += n["refs"]
        for v in n.values():
            _refs(v, acc)
    elif isinstance(n, list):
        for v in n:
            _refs(v, acc)
    return acc


def _concat_operands(st):
    a0 = st["args"][0]
    if "refs" in a0:
        return list(a0["refs"])
    if "list" in a0 and all(isinstance(v, dict) and "ref" in v for v in a0["list"]):
        return [v["ref"] for v in a0["list"]]
    return None


def _lit_int(node):
    return node["int"] if isinstance(node, dict) and "int" in node else None


def fold_transposed_splits(sts, shapes, outputs, multi):
    """`U = concat([reshape(A_l, [N, C, P_l]) ...], axis = 2)`; `T = transpose(U, [0, 2, 1])`; `H_k = split(T, axis = 2, sizes)` -- the
    tail of lele's detection heads (yolo26seg.rs: three passes over [N, 116, 8400], each a copy) -> one transposing copy per
    (level l, head k): channels [c0_k, c1_k) of A_l straight into rows [p_l, p_l + P_l) of H_k (`transpose_cp`, a `window` of a
    reserved [N, P, c_k] buffer).  The same values in the same places; nothing else reads U, T or the reshapes.  Returns
    (statements, number of tails rewritten)."""
    prod, readers = {}, {}
    for i, st in enumerate(sts):
        for o in st.get("out", []):
            prod[o] = i
        names = _refs(st.get("args", st.get("in")), [])
        if st["op"] == "if":
            names += _refs([st.get("then"), st.get("else"), st.get("cond")], [])
        for r in names:
            readers.setdefault(r, []).append(i)

    def call(name, fn):
        i = prod.get(name)
        return i if i is not None and name not in multi and name not in outputs and sts[i]["op"] == "call" and sts[i].get("fn") == fn \
            and len(sts[i]["out"]) == 1 else None

    def ints(node):
        vals = node.get("list") if isinstance(node, dict) else None
        return None if vals is None or any(_lit_int(v) is None for v in vals) else [_lit_int(v) for v in vals]

    drop, put, count = set(), {}, 0
    for i, st in enumerate(sts):
        if st["op"] != "call" or st.get("fn") != "split" or not isinstance(st["args"][0], dict) or "ref" not in st["args"][0]:
            continue
        t = st["args"][0]["ref"]
        sizes = ints(st["args"][2])
        ts = shapes.get(t)
        if sizes is None or ts is None or len(ts) != 3 or _lit_int(st["args"][1]) not in (2, -1) or sum(sizes) != int(ts[2]) \
                or len(sizes) != len(st["out"]) or any(o in multi for o in st["out"]) or readers.get(t) != [i]:
            continue
        it = call(t, "transpose")
        if it is None or ints(sts[it]["args"][1]) != [0, 2, 1] or "ref" not in sts[it]["args"][0]:
            continue
        u = sts[it]["args"][0]["ref"]
        iu = call(u, "concat")
        us = shapes.get(u)
        if us is not None and len(us) == 3 and readers.get(u) == [it] and u not in multi and \
                (iu is None or _lit_int(sts[iu]["args"][1]) in (1, -2)):
            # form B (lele's own Yolo26n-seg tail): U [N, C, P] of any origin -> Transpose -> Split: head k is the transposing copy of
            # channels [c0_k, c1_k) of U -- or, when U is a Concat along C whose operands are exactly the heads, of operand k itself
            # (the Concat then has no reader left and goes too)
            ops = _concat_operands(sts[iu]) if iu is not None else None
            direct = ops is not None and len(ops) == len(sizes) and all(
                shapes.get(o) is not None and len(shapes[o]) == 3 and int(shapes[o][1]) == ck and o not in multi for o, ck in zip(ops, sizes))
            def f32_of(name, depth=0):   # produced by an f32 operator, or a re-arrangement of such values
                pi = prod.get(name)
                if pi is None or name in multi or depth > 16 or sts[pi]["op"] != "call":
                    return False
                fn = sts[pi].get("fn")
                if fn in _F32_FNS:
                    return True
                if fn in ("reshape", "flatten", "unsqueeze", "squeeze", "identity", "transpose", "concat"):
                    srcs = _concat_operands(sts[pi]) if fn == "concat" else [sts[pi]["args"][0].get("ref")]
                    return bool(srcs) and all(o is not None and f32_of(o, depth + 1) for o in srcs)
                return False
            if all(f32_of(o) for o in ops) if direct else f32_of(u):
                new, c0 = [], 0
                for k, (o, ck) in enumerate(zip(st["out"], sizes)):
                    if direct:
                        new.append({"op": "call", "out": [o], "fn": "transpose_cp", "args": [{"ref": ops[k]}], "bufs": 1})
                    else:
                        src = "%s__c" % o
                        new.append({"op": "chview", "out": [src], "src": u, "c0": c0, "c1": c0 + ck})
                        new.append({"op": "call", "out": [o], "fn": "transpose_cp", "args": [{"ref": src}], "bufs": 1})
                    c0 += ck
                put[i] = new
                drop |= {it} | ({iu} if direct else set())
                count += 1
                continue
        if iu is None or readers.get(u) != [it] or _lit_int(sts[iu]["args"][1]) not in (2, -1):
            continue
        levels, ok = [], True
        for lv in _concat_operands(sts[iu]) or [None]:
            il = call(lv, "reshape") if lv else None
            if il is None or readers.get(lv) != [iu] or "ref" not in sts[il]["args"][0]:
                ok = False
                break
            a = sts[il]["args"][0]["ref"]
            sa, sl = shapes.get(a), shapes.get(lv)
            pa = prod.get(a)
            f32 = pa is not None and a not in multi and (sts[pa].get("fn") in _F32_FNS or (sts[pa].get("fn") == "concat" and all(
                prod.get(o) is not None and sts[prod[o]].get("fn") in _F32_FNS for o in (_concat_operands(sts[pa]) or [None]))))
            if not f32 or sa is None or sl is None or len(sa) != 4 or len(sl) != 3 or list(sl[:2]) != list(sa[:2]) or int(sl[2]) != int(sa[2]) * int(sa[3]) \
                    or int(sl[1]) != int(ts[2]) or int(sl[0]) != int(ts[0]):
                ok = False
                break
            levels.append((a, int(sl[2]), il))
        if not ok or not levels or sum(p for _a, p, _i in levels) != int(ts[1]):
            continue
        new, after = [], []
        n = int(ts[0])
        c0 = 0
        for k, (o, ck) in enumerate(zip(st["out"], sizes)):
            buf = o + "__rows"
            new.append({"op": "reserve", "out": [buf], "shape": [n, int(ts[1]), ck], "bufs": 1})
            p0 = 0
            for li, (a, pl, _il) in enumerate(levels):
                src = "%s__l%d" % (o, li)
                new.append({"op": "chview", "out": [src], "src": a, "c0": c0, "c1": c0 + ck})
                piece = "%s__t%d" % (o, li)
                new.append({"op": "call", "out": [piece], "fn": "transpose_cp", "args": [{"ref": src}], "window": {"of": buf, "c0": p0}, "bufs": 0})
                after.append(piece)
                p0 += pl
            new.append({"op": "chview", "out": [o], "src": buf, "c0": 0, "c1": int(ts[1]), "after": [x for x in after if x.startswith(o + "__t")]})
            c0 += ck
        put[i] = new
        drop |= {it, iu} | {il for _a, _p, il in levels}
        count += 1
    out = []
    for i, st in enumerate(sts):
        if i in put:
            out += put[i]
        elif i not in drop:
            out.append(st)
    return out, count


def fold_channel_views(plan, shapes, residuals=True):
    """Returns a format-3 plan in which (after fuse_residual_adds, unless residuals=False), wherever every party can work on a channel view,
      * a Split along C of a rank-4 tensor is a set of views of its operand (no kernel),
      * the producers of a Concat's operands write straight into the Concat's buffer (a `reserve` statement sizes it before the
        first of them runs; operands that cannot be produced in place are copied in by `copy_view`) and the Concat itself is a view,
      * the results of a Split that are, complete and in order, operands of a Concat are handled as ONE operand (the Split's input).
    `shapes`: value name -> shape, recorded by an eager run of `plan` (Runner.shapes).  Anything the pass cannot prove stays as it is."""
    from .compiler.lower import allocate
    import copy
    sts = copy.deepcopy(plan["statements"])
    prod, multi = {}, set()
    for i, st in enumerate(sts):
        for o in st.get("out", []):
            if o in prod:
                multi.add(o)
            prod[o] = i
    n_res = n_tails = 0
    if residuals:
        sts, n_res = fuse_residual_adds(sts, shapes, set(plan["outputs"]), multi)
        if os.environ.get("LELE_AMD_FOLD_TAILS", "1") != "0":   # (0: keep Concat -> Transpose -> Split as three passes, for A/B timing)
            sts, n_tails = fold_transposed_splits(sts, shapes, set(plan["outputs"]), multi)
    nst = len(sts)
    prod = {}
    for i, st in enumerate(sts):
        for o in st.get("out", []):
            prod[o] = i
    readers = {}
    for i, st in enumerate(sts):
        names = _refs(st.get("args", st.get("in")), [])
        if st["op"] == "if":   # whatever a branch reads counts as a reader that cannot take a view
            names += _refs([st.get("then"), st.get("else"), st.get("cond")], [])
        for pos, r in enumerate(names):
            readers.setdefault(r, []).append(i)
    outputs = set(plan["outputs"])

    def rank4(name):
        s_ = shapes.get(name)
        return s_ is not None and len(s_) == 4 and all(int(d) > 0 for d in s_)

    f32_memo = {}

    def is_f32(name, depth=0):
        if name in f32_memo:
            return f32_memo[name]
        ok = False
        if name in prod and name not in multi and depth < 64:
            st = sts[prod[name]]
            fn = st.get("fn")
            if fn in _F32_FNS:
                ok = True
            elif fn == "split":
                ok = is_f32(st["args"][0].get("ref"), depth + 1)
            elif fn == "concat":
                ops = _concat_operands(st)
                ok = bool(ops) and all(is_f32(o, depth + 1) for o in ops)
        elif name in plan["inputs"]:
            info = {d["name"]: d for d in plan.get("input_info", [])}
            ok = info.get(name, {}).get("dtype", "f32") == "f32"
        f32_memo[name] = ok
        return ok

    def axis1(st, pos, name):
        ax = _lit_int(st["args"][pos])
        return ax is not None and rank4(name) and (ax == 1 or ax == -3)

    cand_concat = {}   # statement index -> operand names
    for i, st in enumerate(sts):
        if st["op"] == "call" and st.get("fn") == "concat" and len(st["out"]) == 1:
            ops = _concat_operands(st)
            r = st["out"][0]
            if ops and r not in multi and axis1(st, 1, r) and all(rank4(o) and o not in multi and is_f32(o) for o in ops) \
                    and all(list(shapes[o][:1]) + list(shapes[o][2:]) == list(shapes[r][:1]) + list(shapes[r][2:]) for o in ops) \
                    and sum(int(shapes[o][1]) for o in ops) == int(shapes[r][1]) and len(set(ops)) == len(ops):
                cand_concat[i] = ops
    view_split = {}    # statement index -> True, decided in reverse program order (a split may read another split's view)

    def reader_ok(si, name):
        """can statement si read `name` as a channel view?"""
        st = sts[si]
        if st["op"] != "call":
            return False
        fn, args = st.get("fn"), st.get("args", [])
        where = [k for k, a in enumerate(args) if isinstance(a, dict) and a.get("ref") == name]
        if fn == "conv2d_res":   # the input and the residual may be views (x_pitch / y_pitch)
            return all(k in (0, 3) for k in where) and _conv_group(st) == 1
        if fn in _CONVS:
            return where == [0] and _conv_group(st) == 1
        if fn in ("add", "sub", "mul", "div"):
            other = [a.get("ref") for a in args[:2] if isinstance(a, dict)]
            return len(args) >= 2 and all(k in (0, 1) for k in where) and all(o is not None and shapes.get(o) == shapes.get(name) for o in other) \
                and rank4(name) and is_f32(name)
        if fn in ("max_pool2d", "resize_nearest"):
            return where == [0]
        if fn == "concat":
            return si in cand_concat
        if fn == "split":
            return view_split.get(si, False)
        return False

    for i in range(nst - 1, -1, -1):
        st = sts[i]
        if st["op"] == "call" and st.get("fn") == "split" and "ref" in st["args"][0]:
            x = st["args"][0]["ref"]
            sizes = st["args"][2].get("list") if isinstance(st["args"][2], dict) else None
            if sizes is None or any(_lit_int(v) is None for v in sizes) or not axis1(st, 1, x) or not is_f32(x) or x in multi:
                continue
            if any(o in multi or o in outputs for o in st["out"]):
                continue
            if sum(_lit_int(v) for v in sizes) != int(shapes[x][1]) or len(sizes) != len(st["out"]):
                continue
            view_split[i] = all(reader_ok(si, o) for o in st["out"] for si in readers.get(o, []))
    view_split = {i: v for i, v in view_split.items() if v}

    # ---- concat operands: merge complete, ordered runs of one view-split's results into that split's input; pick the in-place ones
    windowed = {}       # producing statement index -> (concat buffer name, channel offset)
    reserve_at = {}     # statement index before which a `reserve` goes -> [(buffer name, shape)]
    concat_plan = {}    # concat statement index -> (buffer name, [(operand, c0, in_place)])
    for ic in sorted(cand_concat):
        ops = cand_concat[ic]
        r = sts[ic]["out"][0]
        merged, k = [], 0
        while k < len(ops):
            o = ops[k]
            pi = prod.get(o)
            if pi is not None and pi in view_split and sts[pi]["out"][0] == o and sts[pi]["out"] == ops[k:k + len(sts[pi]["out"])]:
                merged.append(sts[pi]["args"][0]["ref"])
                k += len(sts[pi]["out"])
            else:
                merged.append(o)
                k += 1
        buf = r + "__cat"
        entries, c0, first = [], 0, None
        for o in merged:
            pi = prod.get(o)
            ok = pi is not None and pi < ic and pi not in windowed and o not in multi and o not in outputs and sts[pi]["op"] == "call" \
                and len(sts[pi]["out"]) == 1 and sts[pi].get("fn") in _VIEW_WRITERS and "window" not in sts[pi]
            if ok:
                pst = sts[pi]
                if pst["fn"] in _CONVS:
                    ok = _conv_group(pst) == 1
                elif pst["fn"] in ("add", "sub", "mul", "div"):
                    ab = [a.get("ref") if isinstance(a, dict) else None for a in pst["args"][:2]]
                    ok = all(n is not None and shapes.get(n) == shapes.get(o) for n in ab)
            if ok:   # everybody else who reads the operand must cope with a view
                ok = all(si == ic or reader_ok(si, o) for si in readers.get(o, []))
            if ok:
                windowed[pi] = (buf, c0)
                first = pi if first is None else min(first, pi)
            entries.append((o, c0, bool(ok)))
            c0 += int(shapes[o][1])
        if first is None and not any(prod.get(o) in windowed or prod.get(o) in view_split for o, _c, _p in entries) \
                and not any(prod.get(o) in view_split for o in ops):
            continue                      # nothing can be produced in place and every operand is dense: the concat stays ONE copy kernel
        # (the second test looks at the Concat's OWN operands: results of a Split that became views -- because this Concat counted as
        # a reader that copes with views -- must not be left to a plain `concat`, which takes dense tensors only, even when merging them
        # back into the Split's input made every ENTRY dense: y = relu(x); a, b = split(y); z = concat([a, b]))
        concat_plan[ic] = (buf, entries)  # (an operand that is a view elsewhere is copied in by copy_view, which reads views)
        reserve_at.setdefault(ic if first is None else first, []).append((buf, [int(d) for d in shapes[r]]))

    out = []
    for i, st in enumerate(sts):
        for buf, shp in reserve_at.get(i, []):
            out.append({"op": "reserve", "out": [buf], "shape": shp, "bufs": 1})
        if i in windowed:
            buf, c0 = windowed[i]
            st = dict(st, window={"of": buf, "c0": c0}, bufs=0)
            st.pop("slots", None)
            out.append(st)
        elif i in view_split:
            x, c0 = st["args"][0]["ref"], 0
            for o, v in zip(st["out"], st["args"][2]["list"]):
                out.append({"op": "chview", "out": [o], "src": x, "c0": c0, "c1": c0 + v["int"]})
                c0 += v["int"]
        elif i in concat_plan:
            buf, entries = concat_plan[i]
            for o, c0, in_place in entries:
                if not in_place:
                    out.append({"op": "call", "out": [st["out"][0] + "__in%d" % c0], "fn": "copy_view", "args": [{"ref": o}],
                                "window": {"of": buf, "c0": c0}, "bufs": 0})
            # the in-place operands are read here so that liveness keeps the buffer (and them) until the concat's place in the order
            out.append({"op": "chview", "out": [st["out"][0]], "src": buf, "c0": 0, "c1": int(shapes[st["out"][0]][1]),
                        "after": [o for o, _c, _p in entries]})
        else:
            st = dict(st)
            st.pop("slots", None)
            out.append(st)
    slots = allocate(out, list(plan["outputs"]))
    new = dict(plan)
    new.update({"format": "lele_amd.plan/3", "statements": out, "slots": slots,
                "folded": {"residual_adds_fused": n_res, "transposed_splits_folded": n_tails, "concats_in_place": len(concat_plan), "splits_as_views": len(view_split),
                           "operands_in_place": len(windowed), "operands_copied": sum(1 for _b, e in concat_plan.values() for x in e if not x[2])}})
    return new


class Runner:
    def __init__(self, plan, weights, ctx):
        from . import kernels as K
        from ._lib import Weight
        self.plan, self.K, self.ctx = plan, K, ctx
        self.v2 = plan.get("format") in ("lele_amd.plan/2", "lele_amd.plan/3")   # /3: /2 + channel views (fold_channel_views); compiled plans key weights by (offset, kind, shape); lifted ones by offset
        self.raw = {(k if self.v2 else int(k)): v for k, v in weights.items()}
        self.W = {k: (Weight(a) if a.dtype != np.int64 else a) for k, a in self.raw.items()}
        self.ws = {s: ctx.buf() for s in plan["slots"]}
        self.extra = {}
        self.calls = 0
        self.profile = None
        self.stmt_times = None   # set to []: (statement index, fn, first result, device ms) of every kernel statement of the next run
        self.stmt_repeat = 1     # with stmt_times: issue every statement this many times back to back and report the mean -- a single
        #                          eager launch of a 10 us kernel reads as 30 us (launch latency), a train of 8 as its duration
        self.stmt_index = 0
        self.shapes = None  # set to {} to record the shape of every tensor value of the next run
        self.taps = None    # set to {name: None, ...}: host copies of those results are left there by the next run (tests)
        self.n_events = int(plan["dag"]["events"]) if "dag" in plan else 0                   # a DAG plan (lele_amd.lanes.schedule)
        self.event_base = ctx.lane_events(self.n_events) if self.n_events else 0

    def close(self):
        """hand the plan's event ids back to the context (they are a finite space: 65536 per context)"""
        if getattr(self, "n_events", 0):
            self.ctx.lane_events_release(self.event_base, self.n_events)
            self.n_events = 0

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

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
        if fn in ("lstm", "gru", "dynamic_quantize_linear", "fused_quantized_linear_residual_ln", "sanm_out_block", "fused_ffn_quantized_ln"):
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
        try:
            self.exec(self.plan["statements"], env)
        finally:
            if getattr(self.ctx, "cur_lane", 0) != 0:      # a failed statement on a side lane: the context goes back to lane 0 whatever happened
                self.ctx.lane_set(0)
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
            if "lane" in st:     # a DAG plan: this statement's stream, and the points of other lanes it has to wait for
                ctx.lane_set(st["lane"])
                for e in st.get("wait", ()):
                    ctx.lane_wait(self.event_base + e)
            if op == "if":
                self.if_(st, env)
            elif op == "join":   # back on lane 0, after the last statement of every side lane
                ctx.lane_set(0)
                for e in st.get("wait", ()):
                    ctx.lane_wait(self.event_base + e)
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
            elif op == "reserve":   # the buffer of a Concat whose operands are written in place (fold_channel_views)
                from .tensor import TensorView
                from ._lib import DevTensor
                buf = self.ws[st["slots"][0]]
                buf.reserve(4 * int(np.prod(st["shape"], dtype=np.int64)))
                env[st["out"][0]] = TensorView(DevTensor(buf, st["shape"], np.float32))
            elif op == "chview":    # channels [c0, c1) of a device tensor, no copy
                env[st["out"][0]] = env[st["src"]].channels(st["c0"], st["c1"])
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

                    def issue():
                        if "window" in st:   # the result is a channel window of an already reserved tensor (fold_channel_views)
                            whole = env[st["window"]["of"]]
                            d = whole.raw()
                            inner = int(np.prod(d.shape[2:], dtype=np.int64))
                            return f(*pos, out=d.buf, out_window=(d.offset + st["window"]["c0"] * inner, d.pitch or d.shape[1] * inner), ctx=ctx)
                        return self.call(f, fn, pos, bufs, key=st["out"][0] + str(self.stmt_index), may_alias=bool(st.get("may_alias")))

                    res = issue()
                    if self.stmt_times is not None:
                        # per-statement DEVICE time: the statement (a pure function of its operands: the same result again) recorded
                        # `stmt_repeat` times into a hipGraph whose replay is timed with HIP events -- a train issued from here reads as
                        # the host's issue rate for anything shorter than ~40 us (a 20 us convolution as 45), which is what the lane
                        # scheduler would then plan with.  Statements that cannot be recorded (they allocate or synchronise) keep the train.
                        ms = None
                        if self.stmt_repeat > 1:
                            try:
                                ctx.sync()
                                ctx.graph_begin()
                                for _ in range(self.stmt_repeat):
                                    issue()
                                g = ctx.graph_end()
                                g.launch()
                                ctx.sync()
                                ctx.timer_start()
                                g.launch()
                                ms = ctx.timer_stop() / self.stmt_repeat
                                g.close()
                            except Exception:  # noqa: BLE001
                                ctx.graph_abort()   # (a no-op outside capture)
                                ms = None
                        if ms is None:
                            ctx.timer_start()
                            for _ in range(self.stmt_repeat):
                                issue()
                            ms = ctx.timer_stop() / self.stmt_repeat
                        self.stmt_times.append((self.stmt_index, fn, st["out"][0], ms))
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
            if "record" in st:
                ctx.lane_record(self.event_base + st["record"])
