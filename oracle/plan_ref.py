"""oracle/plan_ref.py -- a whole model on the CPU oracle: a statement-by-statement executor of a device plan over `pyoracle` + `npref`.

TEST INFRASTRUCTURE (see oracle/oracle.h): only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

A plan is the call sequence of a lele-generated model -- lifted from generated Rust (tools/lift_generated.py: one statement per
`let y = lele::kernels::<op>(..)` of e.g. /root/reference/examples/yolo26n-seg/src/yolo26seg.rs:300-660) or compiled from ONNX by
lele_amd.compiler (the same statements in this repository's own emitter's words).  The device runs it through the C ABI
(lele_amd/plan.py Runner); this module runs THE SAME STATEMENTS on numpy arrays, each through the oracle's restatement of the
reference kernel of that name, so that a device forward can be held against the reference's arithmetic at the level the reference
itself works at: the graph.  Every operator maps to the function that restates the reference routine:

    conv2d / conv2d_silu / conv2d_fused   pyoracle.conv2d_im2col   src/kernels/conv2d.rs:107-176, 597-760 (+ avx/math.rs epilogues)
    conv_transpose                        pyoracle.conv_transpose  conv2d.rs:2952-3128
    matmul / matmul_fused_add / gemm      pyoracle.*               gemm.rs:112-535 (k-ordered f32 FMA chain; faer's order is unpinned)
    softmax / layer_norm / batch_norm     pyoracle.*               avx/norm.rs:10-283, norm.rs:310-378
    sigmoid / silu / relu / exp / tanh .. pyoracle.unary           avx/math.rs (8-wide polynomial bodies, libm tails)
    add / sub / mul / div / mod_f32 ..    npref.binary             math.rs:69-264, 414-1192 (one IEEE operation per element)
    concat / split / slice / transpose / tile / expand / reshape family / gather / gather_elements / resize_nearest /
    max_pool2d / topk / reduce_*          npref.* / numpy           manipulation.rs, shape.rs, conv2d.rs:1051-1435, math.rs:1527-1920
                                                                    (topk: STABLE descending sort, indices as f32: conv2d.rs:1385-1435)

The forms only this repository's compiler emits are run as the operator sequences they stand for (lele_amd/kernels.py states the
equivalence of each, and tests/test_compiler.py pins it on the device): matmul_view = views + matmul (+ transpose / reshape),
softmax_scaled = mul + softmax, add3 = add(add(a, b), c), view_copy = the chain's slice / reshape / transpose copies,
conv2d_res = conv2d* followed by add.

`run(plan, weights, inputs)` returns the plan's outputs; `taps` collects named intermediate values (the pre-top-k score and box
tensors of a detection head, for instance)."""
import numpy as np

from . import npref
from . import pyoracle as O

_UNARY = {"sigmoid", "silu", "relu", "exp", "tanh", "tanh_kernel", "erf", "gelu", "fast_gelu", "sqrt", "log", "sin", "cos", "neg", "reciprocal",
          "softplus", "not"}
_BINARY = {"add", "sub", "mul", "div", "pow", "max", "min", "equal", "less", "greater", "prelu", "mod_f32"}
_REDUCE = {"reduce_sum": "sum", "reduce_mean": "mean", "reduce_max": "max", "reduce_l2": "l2"}


def _resolve_shape(shape, target):
    """shape.rs:2-52: 0 copies the input dimension, one -1 is inferred"""
    total = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
    out, infer = [], None
    for i, d in enumerate(target):
        d = int(d)
        if d == -1:
            infer = i
            out.append(1)
        elif d == 0:
            out.append(int(shape[i]) if i < len(shape) else 1)
        else:
            out.append(d)
    if infer is not None:
        known = int(np.prod(out, dtype=np.int64))
        out[infer] = 0 if known == 0 else total // known
    return out


def _apply_chain(x, chain):
    """["slice", axis, start, length] / ["reshape", dims] / ["transpose", perm] (lele_amd/kernels.py view_copy): the operators, in order"""
    for step in chain or []:
        if step[0] == "slice":
            _, axis, start, length = step
            sl = [slice(None)] * x.ndim
            sl[axis] = slice(start, start + length)
            x = x[tuple(sl)]
        elif step[0] == "reshape":
            x = np.ascontiguousarray(x).reshape(_resolve_shape(x.shape, step[1]))
        elif step[0] == "transpose":
            x = np.transpose(x, [p + x.ndim if p < 0 else p for p in step[1]])
        else:
            raise ValueError("view chain step %r" % (step[0],))
    return np.ascontiguousarray(x)


def _act_of(fn, pos):
    if fn == "conv2d_silu":
        return "silu"
    if fn == "conv2d_fused":
        return "relu" if pos[7] else None
    return None


class PlanRef:
    def __init__(self, plan, weights):
        self.plan = plan
        self.v2 = plan.get("format") in ("lele_amd.plan/2", "lele_amd.plan/3")
        self.W = {(k if self.v2 else int(k)): np.asarray(v) for k, v in weights.items()}
        self.taps = None        # {name: None}: filled with copies of those values by the next run
        self.calls = 0

    def _wkey(self, node):
        kind, off, _ln, shape = node
        return "%d:%s:%s" % (off, kind, "x".join(map(str, shape))) if self.v2 else node[1]

    def val(self, n, env):
        if "ref" in n:
            return env[n["ref"]]
        if "refs" in n:
            return [env[r] for r in n["refs"]]
        if "weight" in n:
            return self.W[self._wkey(n["weight"])]
        if "weight_scalar" in n:
            return int(self.W[self._wkey(n["weight_scalar"])].reshape(-1)[0])
        if "weight_list" in n:
            a = self.W[self._wkey(n["weight_list"])].reshape(-1)
            return [float(v) for v in a] if a.dtype == np.float32 else [int(v) for v in a]
        if "ints" in n:
            return [int(v) for v in np.asarray(env[n["ints"]]).reshape(-1)]
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

    # ------------------------------------------------------------------------------------------------ one kernel statement
    def call(self, fn, p):
        f32 = lambda a: np.ascontiguousarray(a, np.float32)   # noqa: E731
        if fn in ("conv2d", "conv2d_silu", "conv2d_fused"):
            return O.conv2d_im2col(p[0], p[1], p[2], p[3], p[4], p[5], p[6], _act_of(fn, p))
        if fn == "conv2d_res":     # act(conv(x) + bias) + res: conv2d* then add (plan.fuse_residual_adds)
            y = O.conv2d_im2col(p[0], p[1], p[2], p[4], p[5], p[6], p[7], {0: None, 1: "relu", 2: "silu"}[int(p[8])])
            return npref.binary("add", y, f32(p[3]))
        if fn == "conv1d" or fn == "conv1d_fused":
            return O.conv1d(p[0], p[1], p[2], p[3], p[4], p[5], p[6], bool(p[7]) if len(p) > 7 else False)
        if fn == "conv_transpose":
            return O.conv_transpose(p[0], p[1], p[2], p[3], p[4], p[5], p[6])
        if fn == "matmul":
            return O.matmul(f32(p[0]), f32(p[1]), acc32=False)
        if fn == "matmul_fused_add":
            return O.matmul_fused_add(f32(p[0]), f32(p[1]), f32(p[2]), acc32=False)
        if fn == "gemm":
            return O.gemm(p[0], p[1], p[2], p[3], p[4], bool(p[5]), bool(p[6]), acc32=False)
        if fn == "matmul_view":
            a, b = _apply_chain(f32(p[0]), p[1]), _apply_chain(f32(p[2]), p[3])
            y = O.matmul(a, b, acc32=False)
            if len(p) > 4 and p[4]:
                y = np.ascontiguousarray(np.transpose(y, list(p[4])))
            if len(p) > 5 and p[5] is not None:
                y = y.reshape(_resolve_shape(y.shape, p[5]))
            return y
        if fn == "view_copy":
            return _apply_chain(np.asarray(p[0]), p[1])
        if fn == "softmax":
            return O.softmax(p[0], int(p[1]))
        if fn == "softmax_scaled":
            return O.softmax(npref.binary("mul", f32(p[0]), f32(p[1])), int(p[2]))
        if fn == "layer_norm":
            return O.layer_norm(p[0], p[1], p[2], int(p[3]), float(p[4]))
        if fn == "batch_norm":
            return O.batch_norm(p[0], p[1], p[2], p[3], p[4], float(p[5]))
        if fn in _UNARY:
            name = "tanh" if fn == "tanh_kernel" else fn
            return O.unary(name, f32(p[0])) if name in O.UNARY_SIMD else npref.unary_exact(name, f32(p[0]))
        if fn in _BINARY:
            return npref.binary(fn, np.asarray(p[0]), np.asarray(p[1]))
        if fn == "add3":
            return npref.binary("add", npref.binary("add", f32(p[0]), f32(p[1])), f32(p[2]))
        if fn in _REDUCE:
            return npref.reduce(_REDUCE[fn], f32(p[0]), [int(a) for a in p[1]], bool(p[2]))
        if fn == "concat":
            return np.concatenate([np.asarray(t) for t in p[0]], axis=int(p[1]))
        if fn in ("split", "split_owned"):
            cuts = np.cumsum([int(s) for s in p[2]])[:-1]
            return [np.ascontiguousarray(t) for t in np.split(np.asarray(p[0]), cuts, axis=int(p[1]))]
        if fn == "reshape":
            return np.ascontiguousarray(p[0]).reshape(_resolve_shape(np.shape(p[0]), p[1]))
        if fn == "flatten":
            x = np.asarray(p[0])
            ax = int(p[1]) + x.ndim if int(p[1]) < 0 else int(p[1])
            return x.reshape(int(np.prod(x.shape[:ax], dtype=np.int64)), -1 if x.size else 0)
        if fn == "unsqueeze":
            x = np.asarray(p[0])
            r = x.ndim + len(p[1])
            for a in sorted((int(a) + r if int(a) < 0 else int(a)) for a in p[1]):
                x = np.expand_dims(x, a)
            return x
        if fn == "squeeze":
            x = np.asarray(p[0])
            axes = p[1] if len(p) > 1 and p[1] else None
            return np.squeeze(x, tuple(int(a) for a in axes)) if axes else x.reshape([d for d in x.shape if d != 1])
        if fn in ("identity", "cast"):
            return np.asarray(p[0])
        if fn == "cast_to_f32":
            return np.asarray(p[0]).astype(np.float32)
        if fn == "cast_to_i64":
            return np.trunc(np.asarray(p[0])).astype(np.int64)
        if fn == "transpose":
            x = np.asarray(p[0])
            perm = [int(a) for a in p[1]] or list(range(x.ndim))[::-1]
            return np.ascontiguousarray(np.transpose(x, [a + x.ndim if a < 0 else a for a in perm]))
        if fn == "slice":
            return npref.slice_(np.asarray(p[0]), p[1], p[2], p[3], p[4])
        if fn == "tile":
            return np.tile(np.asarray(p[0]), [int(r) for r in p[1]])
        if fn == "expand":        # math.rs:2168-2247: numpy broadcasting of the input against the target
            x = np.asarray(p[0])
            return np.ascontiguousarray(np.broadcast_to(x, np.broadcast_shapes(x.shape, tuple(int(d) for d in p[1]))))
        if fn == "pad":
            return npref.pad(np.asarray(p[0]), p[1], 0 if p[2] is None else np.asarray(p[2]).reshape(-1)[0], p[3] if len(p) > 3 else "constant")
        if fn == "gather":
            return npref.gather(p[0], p[1], int(p[2]))
        if fn == "gather_elements":
            return npref.gather_elements(p[0], p[1], int(p[2]))
        if fn == "resize_nearest":   # conv2d.rs:1283-1326: sizes win over scales; out = floor(in * scale) in f32
            x = f32(p[0])
            scales, sizes = p[1], p[2]
            if sizes is not None and len(sizes) == 4:
                oh, ow = int(sizes[2]), int(sizes[3])
            else:
                oh = int(np.floor(np.float32(x.shape[2]) * np.float32(scales[2])))
                ow = int(np.floor(np.float32(x.shape[3]) * np.float32(scales[3])))
            return npref.resize_nearest(x, oh, ow, p[3] == "asymmetric")
        if fn == "max_pool2d":
            return npref.max_pool2d(f32(p[0]), p[1], p[2], p[3], p[4], bool(p[5]))
        if fn == "topk":
            assert int(p[2]) in (-1, np.ndim(p[0]) - 1), "topk: last axis only (conv2d.rs:1394)"
            return list(npref.topk(f32(p[0]), int(p[1]), bool(p[3])))
        if fn == "lstm":          # rnn.rs:67: (Y [T, 1, 1, H], H_n [1, 1, H], C_n [1, 1, H]); sequence_lens unused upstream
            return list(O.lstm(p[0], p[1], p[2], p[3], p[5], p[6]))
        if fn == "gru":
            assert bool(p[5]) if len(p) > 5 else True, "gru: the oracle restates the linear_before_reset = 1 formula (rnn.rs:246, ref_gru_step)"
            return list(O.gru(p[0], p[1], p[2], p[3], p[4]))
        if fn == "halves_pow_add_sqrt":   # sqrt(pow(x[lo], e_lo) + pow(x[hi], e_hi)) along `axis`: slice, pow, add, sqrt (lele_amd/kernels.py)
            x, axis = f32(p[0]), int(p[1])
            lo = npref.slice_(x, [int(p[2][0])], [int(p[2][1])], [axis], [1])
            hi = npref.slice_(x, [int(p[3][0])], [int(p[3][1])], [axis], [1])
            s_ = npref.binary("add", npref.binary("pow", lo, f32(p[4])), npref.binary("pow", hi, f32(p[5])))
            return np.sqrt(s_.astype(np.float32))
        # ---- the dynamically quantised linears and the fused forms this repository's compiler builds around them (lower.py fold_*):
        #      each runs as the operator sequence it stands for
        if fn == "fused_quantized_linear":       # quantization.rs:77
            return O.fused_quantized_linear(f32(p[0]), p[1], p[2], p[3], p[4], bool(p[5]))
        if fn in ("fused_quantized_linear_residual", "fused_quantized_linear_residual_ln"):
            y = npref.binary("add", O.fused_quantized_linear(f32(p[0]), p[1], p[2], p[3], p[4], bool(p[5])), f32(p[6]))
            if p[7] is not None:
                y = npref.binary("add", y, f32(p[7]))
            return y if fn.endswith("residual") else [y, O.layer_norm(y, p[8], p[9], -1, float(p[10]))]
        if fn == "depthwise_conv1d_tlc":         # Transpose(0,2,1) -> conv1d(group = C) -> Transpose(0,2,1) [-> Add(input)]
            return self._tlc(f32(p[0]), p[1], p[2], int(p[3]), int(p[4]), bool(p[5]), int(p[6]), bool(p[7]))
        if fn == "sanm_out_block":               # memory block -> projection + Adds -> LayerNorm
            mem = self._tlc(f32(p[6]), p[7], p[8], int(p[10]), int(p[11]), False, int(p[9]), True)
            y = npref.binary("add", O.fused_quantized_linear(f32(p[0]), p[1], p[2], p[3], p[4], bool(p[5])), mem)
            if p[12] is not None:
                y = npref.binary("add", y, f32(p[12]))
            return [y, O.layer_norm(y, p[13], p[14], -1, float(p[15]))]
        if fn in ("fused_ffn_quantized", "fused_ffn_quantized_ln"):
            h = O.fused_quantized_linear(f32(p[0]), p[1], p[2], p[3], p[4], True)
            y = O.fused_quantized_linear(h, p[5], p[6], p[7], p[8], bool(p[9]))
            for r in p[10:12]:
                if r is not None:
                    y = npref.binary("add", y, f32(r))
            return y if fn.endswith("quantized") else [y, O.layer_norm(y, p[12], p[13], -1, float(p[14]))]
        if fn == "attention_view":               # matmul_view -> softmax_scaled -> matmul_view (k-ordered f32 sums: the device replica's order)
            q, k, v = _apply_chain(f32(p[0]), p[1]), _apply_chain(f32(p[2]), p[3]), _apply_chain(f32(p[4]), p[5])
            sc = O.matmul(q, k, acc32=self.attention_acc32)
            if p[6] is not None:
                sc = npref.binary("mul", sc, f32(p[6]))
            y = O.matmul(O.softmax(sc, -1), v, acc32=self.attention_acc32)
            if len(p) > 7 and p[7]:
                y = np.ascontiguousarray(np.transpose(y, list(p[7])))
            if len(p) > 8 and p[8] is not None:
                y = y.reshape(_resolve_shape(y.shape, p[8]))
            return y
        if fn == "where_op":
            return npref.where_op(p[0], p[1], p[2])
        if fn == "clip":
            lo = None if p[1] is None else float(np.asarray(p[1]).reshape(-1)[0])
            hi = None if p[2] is None else float(np.asarray(p[2]).reshape(-1)[0])
            return npref.clip(p[0], lo, hi)
        raise NotImplementedError("oracle/plan_ref.py: no restatement bound to plan function %r" % fn)

    attention_acc32 = True   # the two attention products summed k-ordered in f32 (faer's order is unpinned); False: f64

    @staticmethod
    def _tlc(x, w, bias, pl, pr, relu, x_offset, add_input):
        c = int(np.shape(w)[0])
        v = np.ascontiguousarray(x[..., x_offset:x_offset + c])
        y = O.conv1d(np.ascontiguousarray(v.transpose(0, 2, 1)), w, bias, [1], c, [pl, pr], [1], relu)
        y = np.ascontiguousarray(y.transpose(0, 2, 1))
        return npref.binary("add", y, v) if add_input else y

    # ------------------------------------------------------------------------------------------------ the statement loop
    def run(self, inputs):
        env = {k: np.asarray(v) for k, v in inputs.items()}
        self.calls = 0
        self.exec(self.plan["statements"], env)
        return [np.asarray(env[o]) for o in self.plan["outputs"]]

    def exec(self, statements, env):
        for st in statements:
            op = st["op"]
            if op == "ints":
                env[st["out"][0]] = list(st["value"])
            elif op == "newbuf":
                env[st["out"][0]] = None
            elif op == "swap_remove":   # Vec::swap_remove: take element i, move the last element into its place
                lst = env[st["list"]]
                i = st["index"]
                env[st["out"][0]] = lst[i]
                lst[i] = lst[-1]
                lst.pop()
            elif op == "alias":
                env[st["out"][0]] = env[st["src"]]
            elif op == "host":
                from lele_amd.compiler import hostops   # shape arithmetic of the compiled graph: numpy on the host in both runners
                ins = [None if n is None else (np.asarray(n["const"], np.int64 if n["dtype"] == "i64" else np.float32) if "const" in n else np.asarray(env[n["ref"]]))
                       for n in st["in"]]
                for name, r in zip(st["out"], hostops.evaluate(st["onnx"], ins, st["attrs"])):
                    env[name] = np.asarray(r)
            elif op == "if":
                c = np.asarray(self.val(st["cond"], env)).reshape(-1)
                arm = st["then"] if c.size and c[0] != 0 else st["else"]
                self.exec(arm["statements"], env)
                for name, res in zip(st["out"], arm["results"]):
                    env[name] = np.asarray(res["const"], np.int64 if res.get("dtype") == "i64" else np.float32) if "const" in res else env[res["ref"]]
            elif op == "call":
                pos = [self.val(a, env) for a in st["args"] if not ("slot" in a or "buf" in a)]
                res = self.call(st["fn"], pos)
                self.calls += 1
                if len(st["out"]) == 1:
                    env[st["out"][0]] = res
                else:
                    for name, r in zip(st["out"], res):
                        env[name] = r
            else:
                raise NotImplementedError("oracle/plan_ref.py: statement kind %r (channel-view plans are the device's batch form: run the plan they were folded from)" % op)
            if self.taps is not None:   # whatever kind of statement bound the name (a kernel call, a swap_remove out of a split's list, an alias)
                for name in st.get("out", []):
                    if name in self.taps and isinstance(env.get(name), np.ndarray):
                        self.taps[name] = np.array(env[name], copy=True)


def run(plan, weights, inputs, taps=None):
    r = PlanRef(plan, weights)
    r.taps = taps
    return r.run(inputs)


def calibrate(plan, weights, inputs, target_std=1.0):
    """Synthetic weights that keep a deep graph's activations O(1): one forward on the oracle in which every convolution's weight and
    bias are rescaled so that its result BEFORE the activation has standard deviation `target_std` on `inputs` (data-dependent
    initialisation, layer by layer in execution order).  Seeded N(0, 1/sqrt(fan_in)) weights alone let a 118-convolution SiLU network's
    signal die (the detection scores of such a Yolo26n-seg all sit within 1e-5 of 0.515): a graph-level comparison on that is nearly
    blind.  Returns a new weights dict (same keys)."""
    ref = PlanRef(plan, weights)
    ref.W = {k: np.array(v, copy=True) for k, v in ref.W.items()}
    convs = ("conv2d", "conv2d_silu", "conv2d_fused", "conv_transpose")
    inner = ref.call

    def call(fn, p):
        if fn in convs:
            wkey = next(k for k, v in ref.W.items() if v is p[1])
            pre = inner("conv2d" if fn != "conv_transpose" else fn, list(p[:7]))
            s = float(np.std(pre.astype(np.float64)))
            if s > 0:
                g = np.float32(target_std / s)
                ref.W[wkey] = p[1] = (p[1] * g).astype(np.float32)
                if p[2] is not None:
                    bkey = next(k for k, v in ref.W.items() if v is p[2])
                    ref.W[bkey] = p[2] = (p[2] * g).astype(np.float32)
        return inner(fn, p)
    ref.call = call
    ref.run(inputs)
    return ref.W
