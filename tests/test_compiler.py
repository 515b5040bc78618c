"""ONNX -> device plan compiler (lele_amd/compiler).

CPU: the protobuf reader against its writer AND against torch's exporter output, weights.bin packing (alignment, content
de-duplication: src/compiler/mod.rs:1381-1505), pattern fusion (patterns.rs), liveness allocation invariants
(mod.rs:148-290), host shape arithmetic.  GPU: compiled plans run through the C ABI against torch's CPU result (an
independent second opinion, tolerance 1e-4 relative as for every f32 GEMM-class op) and against the oracle for the
quantised-linear pattern; hipGraph replay of a compiled plan."""
import json

import numpy as np
import pytest

from lele_amd.compiler import CompileError, compile_model, hostops
from lele_amd.compiler import onnx_pb as pb
from tests.onnx_util import export, splice_if

torch = pytest.importorskip("torch")


def fns(plan):
    return [st.get("fn", "host:" + st.get("onnx", "")) for st in plan["statements"]]


class Toy(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(0)
        self.c = torch.nn.Conv2d(3, 8, 3, padding=1)
        self.c2 = torch.nn.Conv2d(8, 8, 1)
        self.ln = torch.nn.LayerNorm(16)
        self.fc = torch.nn.Linear(16, 4)

    def forward(self, x):
        y = self.c(x)
        y = y * torch.sigmoid(y)
        y = torch.relu(self.c2(y))
        y = y.mean(dim=1)
        y = self.ln(y)
        y = y.reshape(y.shape[0], -1, 16)
        return torch.softmax(self.fc(y), -1)


def test_protobuf_roundtrip_and_torch_export():
    g = pb.Graph([pb.Node("Conv", ["x", "w", "b"], ["y"], name="c0", strides=[2, 2], pads=[1, 1, 1, 1], group=1, auto_pad="NOTSET", alpha=0.5),
                  pb.Node("Constant", [], ["c"], value=np.array([1, -2, 3], np.int64))],
                 [pb.ValueInfo("x", pb.FLOAT, [1, 3, "h", 640])], [pb.ValueInfo("y", pb.FLOAT, None)],
                 [pb.Tensor("w", np.arange(24, dtype=np.float32).reshape(2, 3, 2, 2)), pb.Tensor("h", np.array([1, 2], np.float16)),
                  pb.Tensor("u", np.array([[1, 2], [3, 250]], np.uint8)), pb.Tensor("i", np.array([-(2 ** 62), 7], np.int64))])
    m = pb.load(pb.Model(g, opset=13).serialize())
    n = m.graph.node[0]
    assert (n.op_type, n.input, n.output, n.name) == ("Conv", ["x", "w", "b"], ["y"], "c0") and m.opset == 13
    assert n.attr("strides").ints == [2, 2] and n.attr("group").i == 1 and n.attr("auto_pad").s == b"NOTSET" and n.attr("alpha").f == 0.5
    assert m.graph.node[1].attr("value").t.array.tolist() == [1, -2, 3]
    assert m.graph.input[0].shape == [1, 3, "h", 640] and m.graph.output[0].shape is None
    got = {t.name: t.array for t in m.graph.initializer}
    assert got["w"].shape == (2, 3, 2, 2) and got["h"].dtype == np.float16 and got["u"].tolist() == [[1, 2], [3, 250]]
    assert got["i"].tolist() == [-(2 ** 62), 7]
    # a real exporter's bytes
    t = Toy()
    mm = pb.load(export(t, (torch.randn(2, 3, 16, 16),), opset=13))
    assert mm.producer == "pytorch" and mm.opset == 13
    ops = [n.op_type for n in mm.graph.node]
    assert ops[:3] == ["Conv", "Sigmoid", "Mul"] and "Pow" in ops and ops[-1] == "Softmax"
    init = {i.name: i.array for i in mm.graph.initializer}
    assert np.array_equal(init["c.weight"], t.c.weight.detach().numpy()) and np.array_equal(init["ln.bias"], t.ln.bias.detach().numpy())
    assert mm.graph.input[0].shape == [2, 3, 16, 16]


def test_fusion_folding_weights_and_allocation():
    t = Toy()
    for opset in (13, 17):
        plan, blob = compile_model(export(t, (torch.randn(2, 3, 16, 16),), opset=opset), "toy")
        # Conv+Sigmoid+Mul, Conv+Relu, the 9-node LayerNorm (opset 13) / the op (17), MatMul+Add; Shape/Gather/Concat folded into the reshape
        assert fns(plan) == ["conv2d_silu", "conv2d_fused", "reduce_mean", "layer_norm", "reshape", "matmul_fused_add", "softmax"], opset
        rs = [s for s in plan["statements"] if s["fn"] == "reshape"][0]
        assert [v["int"] for v in rs["args"][1]["list"]] == [2, -1, 16]
        for kind, off, ln, shape in plan["weights"].values():   # mod.rs:1418-1424: 16-byte aligned views inside the blob
            assert off % 16 == 0 and off + ln <= len(blob) and kind == "weight_f32" and ln == 4 * int(np.prod(shape))
        w = {tuple(v[3]): np.frombuffer(blob[v[1]:v[1] + v[2]], "<f4").reshape(v[3]) for v in plan["weights"].values()}
        assert np.array_equal(w[(8, 3, 3, 3)], t.c.weight.detach().numpy())
    # identical contents are stored once (mod.rs:1408-1417)
    a = np.arange(12, dtype=np.float32)
    g = pb.Graph([pb.Node("Add", ["x", "a"], ["t"]), pb.Node("Mul", ["t", "b"], ["y"])], [pb.ValueInfo("x", pb.FLOAT, [12])],
                 [pb.ValueInfo("y", pb.FLOAT, [12])], [pb.Tensor("a", a), pb.Tensor("b", a.copy())])
    plan, blob = compile_model(pb.Model(g).serialize())
    assert len(blob) == 48 and len({v[1] for v in plan["weights"].values()}) == 1
    # liveness: a statement never writes a slot it reads, and no slot is reassigned while its value is still needed
    plan, _ = compile_model(export(t, (torch.randn(2, 3, 16, 16),), opset=17))
    live, root = {}, {}
    sts = plan["statements"]

    def refs(n, acc):
        if isinstance(n, dict):
            if isinstance(n.get("ref"), str):
                acc.append(n["ref"])
            for v in n.values():
                refs(v, acc)
        elif isinstance(n, list):
            for v in n:
                refs(v, acc)
        return acc
    last = {}
    for i, st in enumerate(sts):
        for r in refs(st.get("args"), []):
            last[r] = i
    for o in plan["outputs"]:
        last[o] = 10 ** 9
    for i, st in enumerate(sts):
        rd = refs(st.get("args"), [])
        if not st.get("slots"):
            root[st["out"][0]] = root.get(rd[0], rd[0])
            continue
        for name, slot in zip(st["out"], st["slots"]):
            assert all(live.get(root.get(r, r)) != slot for r in rd), (st["fn"], slot)
            for other, s in live.items():
                if s == slot:
                    users = [v for v in last if root.get(v, v) == other]
                    assert all(last[v] < i for v in users), (other, slot, i)
            live[name] = slot


def test_quantised_linear_pattern_and_unsupported_ops():
    rng = np.random.default_rng(0)
    k, n = 32, 24
    w = rng.integers(0, 256, (k, n)).astype(np.uint8)
    init = [pb.Tensor("w", w), pb.Tensor("ws", (rng.random(n) * 0.01 + 0.002).astype(np.float32)), pb.Tensor("wz", np.array(128, np.uint8)),
            pb.Tensor("b", rng.standard_normal(n).astype(np.float32))]
    nodes = [pb.Node("DynamicQuantizeLinear", ["x"], ["q", "s", "z"]), pb.Node("Mul", ["s", "ws"], ["cs"]),
             pb.Node("MatMulInteger", ["q", "w", "z", "wz"], ["mm"]), pb.Node("Cast", ["mm"], ["mmf"], to=1),
             pb.Node("Mul", ["mmf", "cs"], ["dq"]), pb.Node("Add", ["dq", "b"], ["lin"]), pb.Node("Relu", ["lin"], ["y"])]
    g = pb.Graph(nodes, [pb.ValueInfo("x", pb.FLOAT, [5, k])], [pb.ValueInfo("y", pb.FLOAT, [5, n])], init)
    plan, _ = compile_model(pb.Model(g).serialize())
    assert fns(plan) == ["fused_quantized_linear"] and plan["statements"][0]["args"][-1] == {"bool": True}
    g.node, g.output = nodes[:6], [pb.ValueInfo("lin", pb.FLOAT, [5, n])]
    plan, _ = compile_model(pb.Model(g).serialize())
    assert fns(plan) == ["fused_quantized_linear"] and plan["statements"][0]["args"][-1] == {"bool": False}
    # an intermediate with a second reader blocks the fusion (the reference does not check this)
    g.node = nodes[:6] + [pb.Node("Add", ["lin", "dq"], ["y2"])]
    g.output = [pb.ValueInfo("y2", pb.FLOAT, [5, n])]
    plan, _ = compile_model(pb.Model(g).serialize())
    assert "fused_quantized_linear" not in fns(plan) and "dynamic_quantize_linear" in fns(plan) and "mat_mul_integer" in fns(plan)
    with pytest.raises(CompileError, match="NonMaxSuppression"):
        compile_model(pb.Model(pb.Graph([pb.Node("NonMaxSuppression", ["x", "x"], ["y"])], [pb.ValueInfo("x", pb.FLOAT, [1])],
                                        [pb.ValueInfo("y", pb.FLOAT, [1])])).serialize())


def test_host_shape_arithmetic():
    x = np.arange(10)
    assert hostops.slice_(x, [1], [8], [0], [2]).tolist() == [1, 3, 5, 7]
    assert hostops.slice_(x, [-1], [-(2 ** 63)], [0], [-1]).tolist() == x[::-1].tolist()
    assert hostops.slice_(x, [-3], [2 ** 63 - 1]).tolist() == [7, 8, 9]
    assert hostops.evaluate("Gather", [np.array([4, 5, 6]), np.array(-1)], {}) [0] == 6
    assert hostops.evaluate("Div", [np.array([7, -7]), np.array([2, 2])], {})[0].tolist() == [3, -3]      # truncating, as Rust
    assert hostops.evaluate("Unsqueeze", [np.array(5), np.array([0])], {})[0].tolist() == [5]
    assert hostops.evaluate("Reshape", [np.arange(6).reshape(2, 3), np.array([0, -1])], {})[0].shape == (2, 3)
    assert hostops.evaluate("Softmax", [x], {}) is None


# --------------------------------------------------------------------------------------------------- on the device
def run_plan(ctx, plan, blob, inputs):
    from lele_amd.plan import Runner, load_weights_bin
    plan = json.loads(json.dumps(plan))  # plans are plain JSON
    r = Runner(plan, load_weights_bin(plan, blob), ctx)
    return r, r.run(inputs)


def close(got, want, tol=1e-4):
    want = np.asarray(want)
    return got.shape == want.shape and np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())


@pytest.mark.gpu
def test_compiled_toy_matches_torch_and_replays_as_graph(ctx):
    t = Toy()
    x = torch.randn(2, 3, 16, 16, generator=torch.Generator().manual_seed(1))
    want = t(x).detach().numpy()
    for opset in (13, 17):
        plan, blob = compile_model(export(t, (x,), opset=opset))
        xbuf = ctx.buf()
        xd = xbuf.upload(x.numpy())
        from lele_amd.tensor import TensorView
        r, outs = run_plan(ctx, plan, blob, {"x": TensorView(xd)})
        assert close(outs[0].numpy(), want), opset
    # recorded once, replayed: same bits as the eager run
    eager = outs[0].numpy().copy()
    ctx.sync()
    ctx.graph_begin()
    outs = r.run({"x": TensorView(xd)})
    g = ctx.graph_end()
    obuf = r.ws[[st for st in r.plan["statements"] if st["out"] == r.plan["outputs"]][0]["slots"][0]]
    xbuf.upload(np.zeros((2, 3, 16, 16), np.float32))
    g.launch()
    ctx.sync()
    assert not np.array_equal(obuf.to_numpy(eager.shape), eager)  # different input, same graph
    xbuf.upload(x.numpy())
    g.launch()
    ctx.sync()
    assert np.array_equal(obuf.to_numpy(eager.shape), eager)


class Seq(torch.nn.Module):
    def __init__(self):
        super().__init__()
        torch.manual_seed(3)
        self.conv = torch.nn.Conv1d(8, 16, 3, padding=1)
        self.lstm = torch.nn.LSTM(16, 32)
        self.gru = torch.nn.GRU(32, 16)
        self.out = torch.nn.Linear(16, 5)

    def forward(self, x):                      # x: [1, 8, T] with T dynamic
        y = torch.tanh(self.conv(x))           # [1, 16, T]
        y = y.permute(2, 0, 1)                 # [T, 1, 16]
        y, _ = self.lstm(y)
        y, _ = self.gru(y)
        y = y.reshape(-1, y.shape[-1])[1:]     # drop the first frame: Slice with run-time shape arithmetic
        z = self.out(y)
        return torch.cat([z, z * 2.0], dim=0).transpose(0, 1)


@pytest.mark.gpu
def test_compiled_sequence_model_dynamic_length(ctx):
    from lele_amd.tensor import TensorView
    s = Seq()
    data = export(s, (torch.randn(1, 8, 12),), opset=17, dynamic_axes={"x": {2: "t"}})
    plan, blob = compile_model(data)
    assert "lstm" in fns(plan) and "gru" in fns(plan) and any(f.startswith("host:") for f in fns(plan))
    from lele_amd.plan import Runner, load_weights_bin
    r = Runner(plan, load_weights_bin(plan, blob), ctx)
    for T in (12, 7, 31):
        x = torch.randn(1, 8, T, generator=torch.Generator().manual_seed(T))
        want = s(x).detach().numpy()
        got = r.run({"x": TensorView(ctx.buf().upload(x.numpy()))})[0].numpy()
        assert close(got, want), T


@pytest.mark.gpu
def test_compiled_quantised_linear_matches_oracle(ctx):
    from lele_amd.tensor import TensorView
    from oracle import pyoracle as O
    rng = np.random.default_rng(0)
    m, k, n = 37, 64, 48
    w = rng.integers(0, 256, (k, n)).astype(np.uint8)
    ws, b = (rng.random(n) * 0.01 + 0.002).astype(np.float32), rng.standard_normal(n).astype(np.float32)
    init = [pb.Tensor("w", w), pb.Tensor("ws", ws), pb.Tensor("wz", np.array(128, np.uint8)), pb.Tensor("b", b)]
    nodes = [pb.Node("DynamicQuantizeLinear", ["x"], ["q", "s", "z"]), pb.Node("Mul", ["s", "ws"], ["cs"]),
             pb.Node("MatMulInteger", ["q", "w", "z", "wz"], ["mm"]), pb.Node("Cast", ["mm"], ["mmf"], to=1),
             pb.Node("Mul", ["mmf", "cs"], ["dq"]), pb.Node("Add", ["dq", "b"], ["lin"]), pb.Node("Relu", ["lin"], ["y"])]
    g = pb.Graph(nodes, [pb.ValueInfo("x", pb.FLOAT, [m, k])], [pb.ValueInfo("y", pb.FLOAT, [m, n])], init)
    x = rng.standard_normal((m, k)).astype(np.float32)
    want = O.fused_quantized_linear(x, w.astype(np.float32), ws, np.array([128.0], np.float32), b, True)
    plan, blob = compile_model(pb.Model(g).serialize())
    _, outs = run_plan(ctx, plan, blob, {"x": TensorView(ctx.buf().upload(x))})
    assert np.array_equal(outs[0].numpy(), want)       # the fused path is bit-exact with the oracle (test_quant.py)
    # the same graph with the fusion blocked runs node by node: DynamicQuantizeLinear, MatMulInteger, ... -- same value class
    g.node = nodes[:6] + [pb.Node("Relu", ["lin"], ["y"]), pb.Node("Identity", ["dq"], ["aux"])]
    g.output = [pb.ValueInfo("y", pb.FLOAT, [m, n]), pb.ValueInfo("aux", pb.FLOAT, [m, n])]
    plan, blob = compile_model(pb.Model(g).serialize())
    assert "fused_quantized_linear" not in fns(plan)
    _, outs = run_plan(ctx, plan, blob, {"x": TensorView(ctx.buf().upload(x))})
    assert close(outs[0].numpy(), want, 1e-5)


class Vision(torch.nn.Module):
    """YOLO-flavoured: conv+SiLU stem, split / concat bottleneck, max-pool pyramid, nearest x2 up-sampling, transposed conv"""

    def __init__(self):
        super().__init__()
        torch.manual_seed(5)
        self.stem = torch.nn.Conv2d(3, 16, 3, stride=2, padding=1)
        self.a = torch.nn.Conv2d(8, 8, 3, padding=1)
        self.dw = torch.nn.Conv2d(16, 16, 3, padding=1, groups=16)
        self.bn = torch.nn.BatchNorm2d(16)
        self.pool = torch.nn.MaxPool2d(5, 1, 2)
        self.fuse = torch.nn.Conv2d(48, 16, 1)
        self.up = torch.nn.Upsample(scale_factor=2, mode="nearest")
        self.ct = torch.nn.ConvTranspose2d(16, 8, 2, stride=2)
        self.head = torch.nn.Conv2d(8, 4, 1)
        with torch.no_grad():
            self.bn.running_mean.uniform_(-0.2, 0.2)
            self.bn.running_var.uniform_(0.5, 1.5)

    def forward(self, x):
        y = torch.nn.functional.silu(self.stem(x))
        p, q = torch.split(y, 8, dim=1)
        y = torch.cat([p, torch.relu(self.a(q)) + q], dim=1)
        y = torch.nn.functional.hardtanh(self.bn(self.dw(y)), 0.0, 6.0)
        y = self.fuse(torch.cat([y, self.pool(y), self.pool(self.pool(y))], dim=1))
        y = self.up(y)[:, :, 1:-1, ::2]
        y = torch.nn.functional.pad(y, (1, 1, 0, 2))
        y = torch.sigmoid(self.head(self.ct(y)))
        return y, y.mean(dim=(2, 3))


class Attention(torch.nn.Module):
    """SenseVoice-flavoured block: LayerNorm, fused QKV, head split, scaled dot-product attention, GELU feed-forward, embedding"""

    def __init__(self, d=32, heads=4):
        super().__init__()
        torch.manual_seed(9)
        self.d, self.h = d, heads
        self.emb = torch.nn.Embedding(10, d)
        self.ln1, self.ln2 = torch.nn.LayerNorm(d), torch.nn.LayerNorm(d)
        self.qkv, self.proj = torch.nn.Linear(d, 3 * d), torch.nn.Linear(d, d)
        self.f1, self.f2 = torch.nn.Linear(d, 4 * d), torch.nn.Linear(4 * d, d)

    def forward(self, x, ids):                                       # x: [B, T, d], ids: [B, 2] int64
        x = torch.cat([self.emb(ids), x], dim=1)
        b, t, _ = x.shape
        q, k, v = self.qkv(self.ln1(x)).chunk(3, dim=-1)
        sp = lambda z: z.reshape(b, t, self.h, self.d // self.h).transpose(1, 2)  # noqa: E731
        att = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / (self.d // self.h) ** 0.5, dim=-1) @ sp(v)
        x = x + self.proj(att.transpose(1, 2).reshape(b, t, self.d))
        x = x + self.f2(torch.nn.functional.gelu(self.f1(self.ln2(x))))
        return torch.where(x > 0, x, x * 0.1).pow(2.0).sqrt()


@pytest.mark.gpu
def test_compiled_vision_and_attention_blocks(ctx):
    from lele_amd.tensor import TensorView
    v = Vision().eval()
    x = torch.randn(2, 3, 32, 32, generator=torch.Generator().manual_seed(2))
    plan, blob = compile_model(export(v, (x,), opset=13, output_names=("y", "m")))
    f = fns(plan)
    for want_fn in ("conv2d_silu", "conv2d_fused", "split", "concat", "max_pool2d", "resize_nearest", "conv_transpose", "slice", "pad", "clip"):
        assert want_fn in f, (want_fn, f)
    _, outs = run_plan(ctx, plan, blob, {"x": TensorView(ctx.buf().upload(x.numpy()))})
    for got, want in zip(outs, v(x)):
        assert close(got.numpy(), want.detach().numpy()), f
    a = Attention().eval()
    xa = torch.randn(2, 9, 32, generator=torch.Generator().manual_seed(4))
    ids = torch.tensor([[1, 7], [3, 3]])
    for opset in (13, 17):
        plan, blob = compile_model(export(a, (xa, ids), opset=opset, input_names=("x", "ids")))
        f = fns(plan)
        assert f.count("layer_norm") == 2 and "softmax" in f and "gather" in f and "erf" in f, f
        _, outs = run_plan(ctx, plan, blob, {"x": TensorView(ctx.buf().upload(xa.numpy())), "ids": ids.numpy()})
        assert close(outs[0].numpy(), a(xa, ids).detach().numpy()), opset


@pytest.mark.gpu
def test_extra_fused_kernels_are_bit_identical_to_their_sequences(ctx):
    from lele_amd import kernels as K
    rng = np.random.default_rng(21)
    x = (rng.standard_normal((3, 4, 37, 171)) * 4).astype(np.float32)
    s = np.array([0.0883883], np.float32)
    for arr in (x, x[..., :8], x.reshape(3, 4, -1)[..., :600], x.reshape(12, -1)[:, :1500]):
        arr = np.ascontiguousarray(arr)
        assert np.array_equal(K.softmax_scaled(arr, s, -1, ctx=ctx).numpy(), K.softmax(K.mul(arr, s, ctx=ctx), -1, ctx=ctx).numpy()), arr.shape
    a, b, c = (rng.standard_normal((5, 33, 64)).astype(np.float32) for _ in range(3))
    assert np.array_equal(K.add3(a, b, c, ctx=ctx).numpy(), K.add(K.add(a, b, ctx=ctx), c, ctx=ctx).numpy())
    assert np.array_equal(K.add3(a, b[:, :1], c[0, 0], ctx=ctx).numpy(), (a + b[:, :1]) + c[0, 0])        # broadcast operands: two passes
    # spectrum magnitude: Slice Pow Slice Pow Add Sqrt == halves_pow_add_sqrt (Slice's bound rules: open / negative ends)
    big = 9223372036854775807
    for shape, axis, lo, hi, ex in (((2, 258, 7), 1, (0, 129), (129, big), (2.0, 2.0)), ((3, 10, 5, 4), 2, (1, 3), (-2, big), (2.0, 3.0)),
                                    ((6, 40), -1, (0, 20), (20, 40), (2.0, 2.0)), ((4, 9), 0, (0, 2), (2, 4), (1.5, 2.0))):
        z = np.abs(rng.standard_normal(shape)).astype(np.float32) * 3
        e0, e1 = (np.array([v], np.float32) for v in ex)
        seq = K.sqrt(K.add(getattr(K, "pow")(K.slice(z, [lo[0]], [lo[1]], [axis], [1], ctx=ctx), e0, ctx=ctx),
                           getattr(K, "pow")(K.slice(z, [hi[0]], [hi[1]], [axis], [1], ctx=ctx), e1, ctx=ctx), ctx=ctx), ctx=ctx).numpy()
        assert np.array_equal(K.halves_pow_add_sqrt(z, axis, lo, hi, e0, e1, ctx=ctx).numpy(), seq), shape
    with pytest.raises(Exception):
        K.halves_pow_add_sqrt(np.ones((2, 9), np.float32), 1, (0, 4), (4, big), np.array([2.0], np.float32), np.array([2.0], np.float32), ctx=ctx)
    for k, (pl, pr), bias in ((11, (5, 5), False), (3, (1, 1), True), (5, (0, 4), True), (7, (6, 0), False), (11, (0, 0), True)):
        xt = rng.standard_normal((3, 29, 48)).astype(np.float32)
        w = rng.standard_normal((48, 1, k)).astype(np.float32)
        bv = rng.standard_normal(48).astype(np.float32) if bias else None
        seq = K.transpose(K.conv1d(K.transpose(xt, [0, 2, 1], ctx=ctx), w, bv, [1], 48, [pl, pr], [1], ctx=ctx), [0, 2, 1], ctx=ctx).numpy()
        assert np.array_equal(K.depthwise_conv1d_tlc(xt, w, bv, pl, pr, ctx=ctx).numpy(), seq), (k, pl, pr, bias)
        if pl + pr == k - 1:   # the FSMN form: channels read in place from a packed tensor, the block's input added
            packed = rng.standard_normal((3, 29, 112)).astype(np.float32)
            packed[:, :, 40:88] = xt
            got = K.depthwise_conv1d_tlc(packed, w, bv, pl, pr, x_offset=40, add_input=True, ctx=ctx).numpy()
            assert np.array_equal(got, K.add(seq, xt, ctx=ctx).numpy()), (k, pl, pr, bias)
    qkv = rng.standard_normal((3, 41, 1536)).astype(np.float32)
    for start, perm in ((0, [0, 2, 1, 3]), (512, [0, 2, 3, 1]), (1024, [0, 2, 1, 3])):
        chain = [["slice", 2, start, 512], ["reshape", [0, 0, 4, 128]], ["transpose", perm]]
        assert np.array_equal(K.view_copy(qkv, chain, ctx=ctx).numpy(), qkv[:, :, start:start + 512].reshape(3, 41, 4, 128).transpose(perm))
    assert np.array_equal(K.view_copy(qkv, [["slice", -1, 100, 7]], ctx=ctx).numpy(), qkv[..., 100:107])
    # a chain that is not ONE strided view (the reshape after the transpose needs a copy) runs step by step: same values
    assert np.array_equal(K.view_copy(qkv, [["transpose", [0, 2, 1]], ["reshape", [-1]]], ctx=ctx).numpy(), qkv.transpose(0, 2, 1).reshape(-1))
    # matmul on views == copy the views out, matmul, transpose the result: same kernels and tiles, hence the same bits
    for b_, t_ in ((3, 41), (1, 504), (32, 171), (2, 7)):
        qkv = rng.standard_normal((b_, t_, 1536)).astype(np.float32)
        cq = [["slice", 2, 0, 512], ["reshape", [0, 0, 4, 128]], ["transpose", [0, 2, 1, 3]]]
        ck = [["slice", 2, 512, 512], ["reshape", [0, 0, 4, 128]], ["transpose", [0, 2, 3, 1]]]
        cv = [["slice", 2, 1024, 512], ["reshape", [0, 0, 4, 128]], ["transpose", [0, 2, 1, 3]]]
        dq = ctx.buf().upload(qkv)
        from lele_amd.tensor import TensorView
        dqv = TensorView(dq)
        sc_seq = K.matmul(K.view_copy(dqv, cq, ctx=ctx), K.view_copy(dqv, ck, ctx=ctx), ctx=ctx)
        sc = K.matmul_view(dqv, cq, dqv, ck, ctx=ctx)
        assert sc.shape == (b_, 4, t_, t_) and np.array_equal(sc.numpy(), sc_seq.numpy()), (b_, t_)
        pr = K.softmax(sc, -1, ctx=ctx)
        av_seq = K.reshape(K.transpose(K.matmul(pr, K.view_copy(dqv, cv, ctx=ctx), ctx=ctx), [0, 2, 1, 3], ctx=ctx), [b_, t_, 512])
        av = K.matmul_view(pr, [], dqv, cv, out_perm=[0, 2, 1, 3], out_reshape=[0, 0, 512], ctx=ctx)
        assert av.shape == (b_, t_, 512) and np.array_equal(av.numpy(), av_seq.numpy()), (b_, t_)
    a2, b2 = rng.standard_normal((37, 20)).astype(np.float32), rng.standard_normal((50, 20)).astype(np.float32)
    assert np.array_equal(K.matmul_view(a2, [], b2, [["transpose", [1, 0]]], ctx=ctx).numpy(), K.gemm(a2, b2, None, 1.0, 0.0, False, True, ctx=ctx).numpy())
    # an operand view with no unit stride among its last two dimensions: the views are copied out and `matmul` runs (the sequence)
    a4, b4 = rng.standard_normal((4, 6, 8, 10)).astype(np.float32), rng.standard_normal((4, 10, 8, 5)).astype(np.float32)
    assert close(K.matmul_view(a4, [["transpose", [0, 3, 1, 2]]], b4, [], ctx=ctx).numpy(), a4.transpose(0, 3, 1, 2) @ b4)


@pytest.mark.gpu
def test_extra_fusions_leave_a_sensevoice_shaped_model_bit_identical(ctx):
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import sensevoice_graph as S
    from lele_amd.tensor import TensorView
    enc = S.Encoder(ctx, layers=3)
    feats = np.random.default_rng(2).standard_normal((1, 41, 560)).astype(np.float32)
    data = S.encoder_onnx(enc, 1)
    plans = {}
    for extra in (False, True):
        plan, blob = compile_model(data, extra_fusions=extra)
        plans[extra] = fns(plan)
        _, outs = run_plan(ctx, plan, blob, {"feats": TensorView(ctx.buf().upload(feats))})
        plans[extra, "out"] = outs[0].numpy()
    extra = {"attention_view", "sanm_out_block", "fused_ffn_quantized_ln"}
    assert extra <= set(plans[True]) and not extra & set(plans[False])
    # per layer: the q / k / v head views live in the attention statement's loaders (matmul_view -> softmax_scaled -> matmul_view
    # as ONE statement: one launch for a batch, the three-call sequence for a grid this small -- same bits either way here), the
    # FSMN convolution reads v in place and adds it
    # ... as the first residual of the output projection, INSIDE that statement (round 6: the memory block, the projection, its Adds and
    # LayerNorm 2 are one statement -- sanm_out_block; the feed-forward block and the NEXT LayerNorm 1 another -- fused_ffn_quantized_ln)
    assert plans[True].count("attention_view") == 3 and plans[True].count("sanm_out_block") == 3 and "depthwise_conv1d_tlc" not in plans[True]
    assert not {"split", "transpose", "reshape", "matmul", "mul", "view_copy", "add", "add3", "matmul_view", "softmax_scaled"} & set(plans[True])
    device = lambda fs: sum(1 for f in fs if not f.startswith("host:") and f not in ("reshape", "flatten", "squeeze", "unsqueeze", "identity"))  # noqa: E731
    # 4 per layer (qkv, attention, memory block + out projection + Adds + LayerNorm 2, feed-forward block + Add + the next LayerNorm 1)
    # + the first LayerNorm 1, the prompt concat and the CTC linear, against 68 device statements as exported (a Split is one statement,
    # three copies)
    assert plans[True].count("fused_ffn_quantized_ln") == 3 and plans[True].count("layer_norm") == 1
    assert (device(plans[False]), device(plans[True])) == (68, 15)
    assert np.array_equal(plans[True, "out"], plans[False, "out"])
    assert np.array_equal(plans[True, "out"], enc.forward(TensorView(ctx.buf().upload(feats))).numpy())


def misc_ops_model():
    """hand-built ONNX nodes for the operators the exported models above do not contain -> (model bytes, input, idx, output names)"""
    rng = np.random.default_rng(17)
    x = rng.standard_normal((2, 3, 8, 8)).astype(np.float32)
    idx = rng.integers(0, 8, (2, 3, 8, 4)).astype(np.int64)
    i64 = lambda *v: np.array(v, np.int64)  # noqa: E731
    init = [pb.Tensor("ax12", i64(1, 2)), pb.Tensor("pads", i64(0, 0, 1, 2, 0, 0, 2, 1)), pb.Tensor("reps", i64(1, 2, 1, 1)),
            pb.Tensor("k3", i64(3)), pb.Tensor("idx", idx), pb.Tensor("slope", np.array([0.25], np.float32)),
            pb.Tensor("three", np.array([3.0], np.float32)), pb.Tensor("sizes", i64(2, 3, 12, 16)), pb.Tensor("roi", np.zeros(0, np.float32)),
            pb.Tensor("scales_none", np.zeros(0, np.float32)), pb.Tensor("two", np.array([2.0], np.float32)),
            pb.Tensor("eshape", i64(2, 2, 3, 8, 8)), pb.Tensor("lo", np.array(-0.5, np.float32)), pb.Tensor("hi", np.array(0.75, np.float32)),
            pb.Tensor("starts", i64(1, 7)), pb.Tensor("ends", i64(8, 0)), pb.Tensor("axes", i64(2, 3)), pb.Tensor("steps", i64(2, -3))]
    nodes = [pb.Node("ReduceSum", ["x", "ax12"], ["rsum"], keepdims=0), pb.Node("ReduceMax", ["x"], ["rmax"], axes=[3], keepdims=1),
             pb.Node("ReduceL2", ["x"], ["rl2"], axes=[1], keepdims=1), pb.Node("ReduceMean", ["x"], ["rmean"], axes=[0, 2], keepdims=0),
             pb.Node("Pad", ["x", "pads"], ["padr"], mode="reflect"), pb.Node("Pad", ["x", "pads", "two"], ["padc"], mode="constant"),
             pb.Node("Tile", ["x", "reps"], ["tiled"]), pb.Node("Expand", ["x", "eshape"], ["expanded"]),
             pb.Node("TopK", ["x", "k3"], ["tkv", "tki"], axis=-1, largest=1), pb.Node("GatherElements", ["x", "idx"], ["gel"], axis=3),
             pb.Node("PRelu", ["x", "slope"], ["prelu"]), pb.Node("Mod", ["x", "three"], ["mod"], fmod=0),
             pb.Node("Softplus", ["x"], ["sp"]), pb.Node("Pow", ["x", "two"], ["sq"]),
             pb.Node("Resize", ["x", "roi", "scales_none", "sizes"], ["rsz"], mode="nearest", coordinate_transformation_mode="asymmetric",
                     nearest_mode="floor"),
             pb.Node("Shape", ["x"], ["shp"]), pb.Node("ConstantOfShape", ["shp"], ["ones"], value=np.array([1.5], np.float32)),
             pb.Node("Add", ["x", "ones"], ["plus"]), pb.Node("Clip", ["x", "lo", "hi"], ["clipped"]),
             pb.Node("Slice", ["x", "starts", "ends", "axes", "steps"], ["sliced"]), pb.Node("Max", ["x", "two", "three"], ["mx3"]),
             pb.Node("Flatten", ["x"], ["flat"], axis=2), pb.Node("Sub", ["flat", "flat"], ["zero"])]
    outs = ["rsum", "rmax", "rl2", "rmean", "padr", "padc", "tiled", "expanded", "tkv", "tki", "gel", "prelu", "mod", "sp", "sq", "rsz", "plus",
            "clipped", "sliced", "mx3", "zero"]
    g = pb.Graph(nodes, [pb.ValueInfo("x", pb.FLOAT, [2, 3, "h", 8])], [pb.ValueInfo(o, pb.FLOAT, None) for o in outs], init)
    return pb.Model(g, opset=13).serialize(), x, idx, outs


@pytest.mark.gpu
def test_lowering_of_the_remaining_operators_against_numpy(ctx):
    """every result of misc_ops_model is a graph output and is compared with a numpy statement of the ONNX definition (index /
    selection ops exactly, arithmetic to 2e-6)"""
    from lele_amd.tensor import TensorView
    data, x, idx, outs = misc_ops_model()
    plan, blob = compile_model(data)
    _, res = run_plan(ctx, plan, blob, {"x": TensorView(ctx.buf().upload(x))})
    got = {o: r.numpy() for o, r in zip(outs, res)}
    order = np.argsort(-x, axis=-1, kind="stable")[..., :3]
    want = {"rsum": x.sum((1, 2)), "rmax": x.max(3, keepdims=True), "rl2": np.sqrt((x.astype(np.float64) ** 2).sum(1, keepdims=True)),
            # reflect follows the reference's own index rule at the far edge (manipulation.rs:382-587; oracle npref.pad), not numpy's
            "rmean": x.mean((0, 2)), "padr": __import__("oracle.npref", fromlist=["pad"]).pad(x, [0, 0, 1, 2, 0, 0, 2, 1], 0.0, "reflect"),
            "padc": np.pad(x, ((0, 0), (0, 0), (1, 2), (2, 1)), constant_values=2.0), "tiled": np.tile(x, (1, 2, 1, 1)),
            "expanded": np.broadcast_to(x, (2, 2, 3, 8, 8)), "tkv": np.take_along_axis(x, order, -1), "tki": order.astype(np.float32),
            "gel": np.take_along_axis(x, idx, 3), "prelu": np.where(x < 0, x * 0.25, x), "mod": x - 3.0 * np.floor(x / 3.0),
            "sp": np.log1p(np.exp(x.astype(np.float64))), "sq": x.astype(np.float64) ** 2,
            "rsz": x[:, :, (np.arange(12) * 8 // 12)][:, :, :, (np.arange(16) * 8 // 16)], "plus": x + 1.5, "clipped": np.clip(x, -0.5, 0.75),
            "sliced": x[:, :, 1:8:2, 7:0:-3], "mx3": np.maximum(np.maximum(x, 2.0), 3.0), "zero": np.zeros((6, 64), np.float32)}
    exact = {"rmax", "padr", "padc", "tiled", "expanded", "tkv", "tki", "gel", "prelu", "rsz", "plus", "clipped", "sliced", "mx3", "zero"}
    for o in outs:
        w = np.asarray(want[o])
        assert got[o].shape == w.shape, (o, got[o].shape, w.shape)
        if o in exact:
            assert np.array_equal(got[o], w.astype(np.float32)), o
        else:
            assert np.abs(got[o] - w).max() <= 2e-6 * max(1.0, np.abs(w).max()), (o, np.abs(got[o] - w).max())


# --------------------------------------------------------------------------------------------------- control flow: If
def if_model():
    """x f32 [2,8], flag i64 [1] -> z.  h = relu(x); if flag == 1 {y = h*w + x; n = 3} else {y = h*sigmoid(h); n = 2};
    z = tile(y + h, [1, n]).  The branches read h and x from the enclosing scope; n is a host integer result."""
    w = np.linspace(-1.0, 1.0, 8).astype(np.float32)
    i64 = lambda *v: np.array(v, np.int64)  # noqa: E731
    then = pb.Graph([pb.Node("Mul", ["h", "w"], ["t1"]), pb.Node("Add", ["t1", "x"], ["y_then"]), pb.Node("Constant", [], ["n_then"], value=i64(3))],
                    [], [pb.ValueInfo("y_then", pb.FLOAT, None), pb.ValueInfo("n_then", pb.INT64, None)], [], "then")
    other = pb.Graph([pb.Node("Sigmoid", ["h"], ["e1"]), pb.Node("Mul", ["h", "e1"], ["y_else"])],
                     [], [pb.ValueInfo("y_else", pb.FLOAT, None), pb.ValueInfo("n_else", pb.INT64, None)], [pb.Tensor("n_else", i64(2))], "else")
    nodes = [pb.Node("Relu", ["x"], ["h"]), pb.Node("Equal", ["flag", "one"], ["cond"]),
             pb.Node("If", ["cond"], ["y", "n"], then_branch=then, else_branch=other),
             pb.Node("Add", ["y", "h"], ["s"]), pb.Node("Concat", ["one", "n"], ["reps"], axis=0), pb.Node("Tile", ["s", "reps"], ["z"])]
    g = pb.Graph(nodes, [pb.ValueInfo("x", pb.FLOAT, [2, 8]), pb.ValueInfo("flag", pb.INT64, [1])], [pb.ValueInfo("z", pb.FLOAT, None)],
                 [pb.Tensor("w", w), pb.Tensor("one", i64(1))])
    return pb.Model(g, opset=13).serialize(), w


def if_reference(x, w, flag):
    h = np.maximum(x, 0.0)
    if flag == 1:
        return np.tile(h * w + x + h, (1, 3))
    return None   # the SiLU branch goes through the device's polynomial sigmoid: compared with the unfused device ops instead


def test_yolo_shaped_graph_compiles_with_the_batch_in_the_graph():
    """tools/yolo_graph.py (BASELINE configs[4] as one graph): the Yolo26n-seg-shaped ONNX compiles on the CPU at N = 1 and N = 64
    to the same call sequence -- every Conv + Sigmoid + Mul folded into conv2d_silu, the attention block into view products, the
    top-300 tail kept (2 topk, 3 gather_elements) -- with the batch size only in the shapes"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from yolo_graph import yolo_onnx
    from lele_amd.compiler import compile_model
    hist = {}
    for n in (1, 64):
        data, info = yolo_onnx(n, 64)
        assert info["convolutions"] == 100
        plan, blob = compile_model(data, "yolo_n%d" % n)
        fns = [s.get("fn") for s in plan["statements"]]
        hist[n] = {f: fns.count(f) for f in set(fns)}
        assert hist[n]["conv2d_silu"] == 86 and hist[n]["conv2d"] == 13 and hist[n]["conv_transpose"] == 1
        assert hist[n]["topk"] == 2 and hist[n]["gather_elements"] == 3 and hist[n]["matmul_view"] == 2 and "sigmoid" in hist[n]
        assert len(blob) > 10_000_000 and "silu" not in hist[n]
    assert hist[1] == hist[64]


def test_if_lowering_structure_and_static_inlining():
    data, _w = if_model()
    m = pb.load(data)
    assert pb.load(m.serialize()).graph.node[2].attribute[0].g.name == "then"     # graph attributes survive a round trip
    plan, _blob = compile_model(data)
    st = [s for s in plan["statements"] if s["op"] == "if"]
    assert len(st) == 1
    st = st[0]
    assert st["out"] == ["y", "n"] and st["kinds"] == ["dev", "host"] and st["dev_out"] == ["y"] and len(st["slots"]) == 1
    assert st["cond"] == {"ref": "cond"} and {"ref": "h"} in st["args"] and {"ref": "x"} in st["args"]
    assert [s["fn"] for s in st["then"]["statements"]] == ["mul", "add"] and st["then"]["results"][1] == {"const": [3], "dtype": "i64"}
    assert [s["fn"] for s in st["else"]["statements"]] == ["silu"] and st["else"]["results"] == [{"ref": "y_else"}, {"const": [2], "dtype": "i64"}]
    inner = {s for arm in ("then", "else") for t in st[arm]["statements"] for s in t["slots"]}
    assert inner and inner <= set(plan["slots"]) and not inner & {s for t in plan["statements"] for s in t.get("slots", [])}
    # h is read inside the branches: its buffer must outlive the `if`, and the `if` must not write into it
    relu = next(s for s in plan["statements"] if s.get("fn") == "relu")
    assert relu["slots"][0] not in st["slots"]
    # the condition known at compile time (the input bound): the taken branch is inlined, no `if` is left
    for flag, want in ((1, ["relu", "mul", "add", "identity", "add", "tile"]), (0, ["relu", "silu", "identity", "add", "tile"])):
        p, _ = compile_model(data, bind={"flag": np.array([flag], np.int64)})
        assert [s.get("fn", s["op"]) for s in p["statements"]] == want and p["inputs"] == ["x"], (flag, fns(p))
    with pytest.raises(CompileError):
        compile_model(data, bind={"nope": 1})


@pytest.mark.gpu
def test_if_runs_both_ways_python_and_native(ctx, tmp_path):
    from lele_amd import kernels as K
    from lele_amd.tensor import TensorView
    from tests.test_native_runner import _native
    data, w = if_model()
    plan, blob = compile_model(data)
    x = np.random.default_rng(3).standard_normal((2, 8)).astype(np.float32)
    xd = TensorView(ctx.buf().upload(x))
    h = np.maximum(x, 0.0)
    silu = K.silu(h, ctx=ctx).numpy()
    want = {1: if_reference(x, w, 1), 0: np.tile(silu + h, (1, 2))}
    for flag in (1, 0, 1):
        _r, res = run_plan(ctx, plan, blob, {"x": xd, "flag": np.array([flag], np.int64)})
        assert np.array_equal(res[0].numpy(), want[flag]), flag
        d = tmp_path / ("f%d_%d" % (flag, len(list(tmp_path.iterdir()))))
        d.mkdir()
        _rec, got = _native(d, plan, blob, {"x": x, "flag": np.array([flag], np.int64)})
        assert np.array_equal(got[0], want[flag]), flag
        p2, b2 = compile_model(data, bind={"flag": np.array([flag], np.int64)})     # the statically inlined form: same bits
        assert np.array_equal(run_plan(ctx, p2, b2, {"x": xd})[1][0].numpy(), want[flag])


class VadNet(torch.nn.Module):
    """Silero-shaped: framing convolution (the STFT basis), magnitude, a small conv encoder, one LSTM step, a sigmoid head"""

    def __init__(self, hop, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.stft = torch.nn.Conv1d(1, 34, 4 * hop, stride=hop, bias=False)
        self.enc1 = torch.nn.Conv1d(17, 32, 3, padding=1)
        self.enc2 = torch.nn.Conv1d(32, 24, 3, stride=2, padding=1)
        self.lstm = torch.nn.LSTM(24, 24)
        self.head = torch.nn.Conv1d(24, 1, 1)

    def forward(self, x, h0, c0):                 # x [1, N], state [1, 1, 24] each
        s = self.stft(x.unsqueeze(1))              # [1, 34, F]
        mag = torch.sqrt(s[:, :17] ** 2 + s[:, 17:] ** 2)
        y = torch.relu(self.enc2(torch.relu(self.enc1(mag))))     # [1, 24, F/2]
        y, (hn, cn) = self.lstm(y.permute(2, 0, 1), (h0, c0))      # [F/2, 1, 24]
        p = torch.sigmoid(self.head(torch.relu(y).permute(1, 2, 0)))
        return p.mean(dim=2), hn, cn


def vad_if_model():
    nets = VadNet(32, 5).eval(), VadNet(16, 6).eval()         # "16 kHz" and "8 kHz" networks
    ex = (torch.zeros(1, 512), torch.zeros(1, 1, 24), torch.zeros(1, 1, 24))
    kw = dict(opset=17, input_names=("x", "h0", "c0"), output_names=("prob", "hn", "cn"))
    parts = [export(n, ex, **kw) for n in nets]
    i64 = lambda *v: np.array(v, np.int64)  # noqa: E731
    data = splice_if(parts[0], parts[1], {"x", "h0", "c0"}, [pb.Node("Equal", ["sr", "sr16k"], ["is16k"])], "is16k",
                     [pb.ValueInfo("x", pb.FLOAT, [1, 512]), pb.ValueInfo("sr", pb.INT64, [1]), pb.ValueInfo("h0", pb.FLOAT, [1, 1, 24]),
                      pb.ValueInfo("c0", pb.FLOAT, [1, 1, 24])], [pb.Tensor("sr16k", i64(16000))])
    return data, nets, parts


def test_silero_shaped_if_compiles_both_networks():
    data, _nets, _parts = vad_if_model()
    plan, blob = compile_model(data, "vad")
    st = [s for s in plan["statements"] if s["op"] == "if"]
    assert len(st) == 1 and st[0]["kinds"] == ["dev", "dev", "dev"] and len(st[0]["slots"]) == 3
    for arm in ("then", "else"):
        f = [s["fn"] for s in st[0][arm]["statements"] if s["op"] == "call"]
        assert f.count("lstm") == 1 and f.count("conv1d_fused") >= 2 and "sigmoid" in f, f
        assert f.count("halves_pow_add_sqrt") == 1 and "pow" not in f and "sqrt" not in f, f   # Slice Pow Slice Pow Add Sqrt -> one call
    assert plan["inputs"] == ["x", "sr", "h0", "c0"] and [i["dtype"] for i in plan["input_info"]] == ["f32", "i64", "f32", "f32"]
    # both networks' weights are in the one weights.bin
    assert len(blob) > 4 * sum(p.numel() for p in _nets[0].parameters()) + 4 * sum(p.numel() for p in _nets[1].parameters()) - 4096
    # sr bound at compile time: only the chosen network is left
    p16, b16 = compile_model(data, "vad16", bind={"sr": np.array([16000], np.int64)})
    assert not [s for s in p16["statements"] if s["op"] == "if"] and len(b16) < 0.75 * len(blob)


@pytest.mark.gpu
def test_silero_shaped_if_matches_torch_per_sample_rate(ctx, tmp_path):
    from lele_amd.tensor import TensorView
    from tests.test_native_runner import _native
    data, nets, parts = vad_if_model()
    plan, blob = compile_model(data, "vad")
    rng = np.random.default_rng(8)
    x = (0.3 * rng.standard_normal((1, 512))).astype(np.float32)
    h0, c0 = (0.1 * rng.standard_normal((2, 1, 1, 24))).astype(np.float32)
    dev = {k: TensorView(ctx.buf().upload(v)) for k, v in (("x", x), ("h0", h0), ("c0", c0))}
    for sr, net, part in ((16000, nets[0], parts[0]), (8000, nets[1], parts[1]), (16000, nets[0], parts[0])):
        with torch.no_grad():
            want = [t.numpy() for t in net(torch.from_numpy(x), torch.from_numpy(h0), torch.from_numpy(c0))]
        _r, res = run_plan(ctx, plan, blob, dict(dev, sr=np.array([sr], np.int64)))
        got = [r.numpy() for r in res]
        for g, w in zip(got, want):
            assert close(g, w), (sr, np.abs(g - w).max())
        # the network compiled on its own (no `If`): the same kernels on the same data, so the same bits
        p1, b1 = compile_model(part, "one")
        alone = [r.numpy() for r in run_plan(ctx, p1, b1, dev)[1]]
        assert all(np.array_equal(a, b) for a, b in zip(got, alone)), sr
        p0, b0 = compile_model(part, "plain", extra_fusions=False)          # lele's own patterns only: the fused forms change no bit
        assert "halves_pow_add_sqrt" not in fns(p0) and "pow" in fns(p0)
        assert all(np.array_equal(a, r.numpy()) for a, r in zip(got, run_plan(ctx, p0, b0, dev)[1])), sr
        d = tmp_path / ("sr%d_%d" % (sr, len(list(tmp_path.iterdir()))))
        d.mkdir()
        _rec, nat = _native(d, plan, blob, {"x": x, "sr": np.array([sr], np.int64), "h0": h0, "c0": c0})
        assert all(np.array_equal(a, b) for a, b in zip(nat, got)), sr


def test_transposes_that_may_be_views_keep_their_operand_alive():
    """a Transpose that only moves size-1 axes is handed on as a view at run time (shapes are dynamic, so the plan cannot know):
    its statement has a slot of its own AND pins the operand's buffer for as long as the result lives"""
    from lele_amd.plan import Runner
    nodes = [pb.Node("Relu", ["x"], ["a"]), pb.Node("Transpose", ["a"], ["t"], perm=[2, 0, 1]), pb.Node("Sigmoid", ["x"], ["b"]),
             pb.Node("Tanh", ["b"], ["c"]), pb.Node("Exp", ["c"], ["d"]), pb.Node("Neg", ["d"], ["e"]), pb.Node("Abs", ["e"], ["f"]),
             pb.Node("Floor", ["f"], ["g"]), pb.Node("Ceil", ["g"], ["h"]), pb.Node("Reshape", ["t", "shp"], ["t2"]), pb.Node("Add", ["t2", "h"], ["y"])]
    g = pb.Graph(nodes, [pb.ValueInfo("x", pb.FLOAT, [1, "c", 1])], [pb.ValueInfo("y", pb.FLOAT, None)], [pb.Tensor("shp", np.array([1, -1, 1], np.int64))])
    plan, _ = compile_model(pb.Model(g, opset=13).serialize())
    st = {s["out"][0]: s for s in plan["statements"]}
    assert st["t"]["may_alias"] and len(st["t"]["slots"]) == 1
    a_slot = st["a"]["slots"][0]
    # a's buffer must not be handed to anything between the transpose and the last reader of t (the Add at the end)
    later = [s for s in plan["statements"][plan["statements"].index(st["t"]):] if s.get("slots")]
    assert all(a_slot not in s["slots"] for s in later), [(s["out"], s["slots"]) for s in later]
    assert Runner.moves_only_unit_axes([1, 128, 1], [2, 0, 1]) and Runner.moves_only_unit_axes([1, 1, 7], [1, 2, 0])
    assert not Runner.moves_only_unit_axes([2, 128, 1], [1, 0, 2]) and Runner.moves_only_unit_axes([5, 1, 6], [0, 2, 1])
    assert not Runner.moves_only_unit_axes([5, 3, 6], [0, 2, 1])


def test_silero_tool_model_compiles_to_the_expected_chain():
    """tools/silero_graph.py (config C1): with sr bound, a chunk is 15 statements of which 12 launch a kernel at run time"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import silero_graph
    data = silero_graph.build_onnx()
    plan, blob = compile_model(data, "silero_shaped", bind={"sr": np.array([16000], np.int64)})
    f = [s["fn"] for s in plan["statements"]]
    assert f == ["unsqueeze", "conv1d", "halves_pow_add_sqrt", "conv1d_fused", "conv1d_fused", "conv1d_fused", "conv1d_fused", "transpose",
                 "lstm", "squeeze", "relu", "transpose", "conv1d", "sigmoid", "reduce_mean", "identity", "identity", "identity"], f
    assert all(s.get("may_alias") for s in plan["statements"] if s["fn"] == "transpose")
    full, blob2 = compile_model(data, "silero_shaped")
    assert [s["op"] for s in full["statements"]] == ["host", "if"] and len(blob2) > 1.7 * len(blob)


@pytest.mark.gpu
def test_extra_fusions_fall_back_where_shapes_rule_the_fused_form_out(ctx):
    """The extra fusions are chosen when the plan is compiled, without shape knowledge: where the run-time shapes rule a
    fused form out, it must run as the node sequence it stands for -- same results as extra_fusions=False, no error.
      (a) a residual that broadcasts OUTWARD after a quantised linear, and an Add -> Add whose last operand enlarges the sum;
      (b) a MatMul of a Reshape -> Transpose view against a rank-2 run-time tensor;
      (c) a Split -> Reshape -> Transpose chain whose Reshape is not expressible in strides (it merges a sliced dimension)."""
    from lele_amd.tensor import TensorView
    rng = np.random.default_rng(11)
    i64 = lambda *v: np.array(v, np.int64)  # noqa: E731

    def both(model, inputs):
        outs = {}
        for extra in (False, True):
            plan, blob = compile_model(model, extra_fusions=extra)
            _, res = run_plan(ctx, plan, blob, {k: TensorView(ctx.buf().upload(v)) for k, v in inputs.items()})
            outs[extra] = [r.numpy() for r in res]
            outs[extra, "fns"] = fns(plan)
        for a, b in zip(outs[False], outs[True]):
            assert a.shape == b.shape and np.array_equal(a, b)
        return outs

    # (a) quantised linear [1, T, N] + residual [3, T, N]; then (x + y) + z with z [2, 3, T, N]
    k, n, t = 32, 24, 5
    w = np.clip(np.round(128 + 32 * rng.standard_normal((k, n))), 0, 255).astype(np.uint8)
    g = pb.Graph([], [pb.ValueInfo("x", pb.FLOAT, [1, t, k]), pb.ValueInfo("r", pb.FLOAT, [3, t, n]), pb.ValueInfo("z", pb.FLOAT, [2, 3, t, n])],
                 [pb.ValueInfo("y", pb.FLOAT, None)],
                 [pb.Tensor("w", w), pb.Tensor("ws", (np.abs(rng.standard_normal(n)) * 0.01 + 0.002).astype(np.float32)), pb.Tensor("wz", np.array(128, np.uint8)),
                  pb.Tensor("b", (rng.standard_normal(n) * 0.02).astype(np.float32))])
    g.node += [pb.Node("DynamicQuantizeLinear", ["x"], ["q", "s", "zp"]), pb.Node("Mul", ["s", "ws"], ["cs"]),
               pb.Node("MatMulInteger", ["q", "w", "zp", "wz"], ["mm"]), pb.Node("Cast", ["mm"], ["mmf"], to=1), pb.Node("Mul", ["mmf", "cs"], ["dq"]),
               pb.Node("Add", ["dq", "b"], ["lin"]), pb.Node("Add", ["lin", "r"], ["s1"]), pb.Node("Add", ["s1", "r"], ["s2"]), pb.Node("Add", ["s2", "z"], ["y"])]
    o = both(pb.Model(g, opset=17).serialize(), {"x": rng.standard_normal((1, t, k)).astype(np.float32), "r": rng.standard_normal((3, t, n)).astype(np.float32),
                                                  "z": rng.standard_normal((2, 3, t, n)).astype(np.float32)})
    assert o[True][0].shape == (2, 3, t, n) and "fused_quantized_linear_residual" in o[True, "fns"]

    # (b) [B, T, H, D] -> transpose -> [B, H, T, D] (a view) x run-time [D, 7]
    g = pb.Graph([], [pb.ValueInfo("x", pb.FLOAT, [2, 6, 32]), pb.ValueInfo("m", pb.FLOAT, [8, 7])], [pb.ValueInfo("y", pb.FLOAT, None)],
                 [pb.Tensor("hs", i64(0, 0, 4, 8))])
    g.node += [pb.Node("Reshape", ["x", "hs"], ["xr"]), pb.Node("Transpose", ["xr"], ["xt"], perm=[0, 2, 1, 3]), pb.Node("MatMul", ["xt", "m"], ["y"])]
    x, m = rng.standard_normal((2, 6, 32)).astype(np.float32), rng.standard_normal((8, 7)).astype(np.float32)
    o = both(pb.Model(g, opset=17).serialize(), {"x": x, "m": m})
    assert close(o[True][0], x.reshape(2, 6, 4, 8).transpose(0, 2, 1, 3) @ m)

    # (c) Split along the last axis, then a Reshape that merges the split axis with the one before it, then Transpose
    g = pb.Graph([], [pb.ValueInfo("x", pb.FLOAT, [2, 6, 12])], [pb.ValueInfo("y", pb.FLOAT, None), pb.ValueInfo("y2", pb.FLOAT, None)],
                 [pb.Tensor("sp", i64(4, 8)), pb.Tensor("rs", i64(2, 3, 8)), pb.Tensor("rs2", i64(2, 6, 2, 4))])
    g.node += [pb.Node("Split", ["x", "sp"], ["a", "b"], axis=2), pb.Node("Reshape", ["a", "rs"], ["ar"]), pb.Node("Transpose", ["ar"], ["y"], perm=[0, 2, 1]),
               pb.Node("Reshape", ["b", "rs2"], ["br"]), pb.Node("Transpose", ["br"], ["y2"], perm=[0, 2, 1, 3])]
    x = rng.standard_normal((2, 6, 12)).astype(np.float32)
    o = both(pb.Model(g, opset=17).serialize(), {"x": x})
    assert np.array_equal(o[True][0], x[:, :, :4].reshape(2, 3, 8).transpose(0, 2, 1))
    assert np.array_equal(o[True][1], x[:, :, 4:].reshape(2, 6, 2, 4).transpose(0, 2, 1, 3))
