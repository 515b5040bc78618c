"""lele_run: the native (C++) plan runner must execute compiled plans exactly like the Python runner -- same statements,
same C-ABI calls, so the outputs are compared bit for bit."""
import json
import os
import subprocess

import numpy as np
import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUN = os.path.join(ROOT, "lele_amd", "lele_run")


def test_runner_binary_builds_and_links():
    from lele_amd import build
    build.build()                      # liblele_hip.so first (cross-compiles without a GPU), then the runner
    assert os.path.exists(RUN)
    r = subprocess.run([RUN], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 2 and "usage: lele_run" in r.stderr


def _native(tmp_path, plan, blob, inputs, extra=()):
    p = tmp_path / "m_plan.json"
    p.write_text(json.dumps(plan))
    (tmp_path / "m_weights.bin").write_bytes(blob)
    cmd = [RUN, str(p), str(tmp_path / "m_weights.bin"), "--out", str(tmp_path / "out")]
    for name, arr in inputs.items():
        f = tmp_path / (name + ".bin")
        arr = np.ascontiguousarray(arr)
        arr.tofile(f)
        cmd += ["--input", "%s=%s:%s:%s" % (name, f, "i64" if arr.dtype == np.int64 else "f32", ",".join(map(str, arr.shape)))]
    r = subprocess.run(cmd + list(extra), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    rec = json.loads(r.stdout.strip().splitlines()[-1])
    outs = [np.fromfile(tmp_path / ("out%d.bin" % k), np.float32).reshape(shape) for k, shape in enumerate(rec["outputs"])]
    return rec, outs


@pytest.mark.gpu
def test_native_runner_matches_python_runner(ctx, tmp_path):
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.tensor import TensorView
    from tests.onnx_util import export
    from tests.test_compiler import Attention, Seq, Toy, Vision
    g = torch.Generator().manual_seed(0)
    cases = [
        ("toy", Toy(), {"x": torch.randn(2, 3, 16, 16, generator=g)}, dict(opset=13)),
        ("seq", Seq(), {"x": torch.randn(1, 8, 23, generator=g)}, dict(opset=17, dynamic_axes={"x": {2: "t"}})),
        ("vision", Vision().eval(), {"x": torch.randn(2, 3, 32, 32, generator=g)}, dict(opset=13, output_names=("y", "m"))),
        ("attention", Attention().eval(), {"x": torch.randn(2, 9, 32, generator=g), "ids": torch.tensor([[1, 7], [3, 3]])},
         dict(opset=17, input_names=("x", "ids"))),
    ]
    from tests.test_compiler import misc_ops_model
    misc_bytes, misc_x, _idx, _outs = misc_ops_model()
    cases.append(("misc", misc_bytes, {"x": torch.from_numpy(misc_x)}, {}))
    for name, model, inp, kw in cases:
        example = tuple(torch.randn(1, 8, 12) if name == "seq" else v for v in inp.values())
        plan, blob = compile_model(model if isinstance(model, bytes) else export(model, example, **kw), name)
        feeds = {k: v.numpy() for k, v in inp.items()}
        r = Runner(plan, load_weights_bin(plan, blob), ctx)
        want = [o.numpy() for o in r.run({k: (TensorView(ctx.buf().upload(v)) if v.dtype != np.int64 else v) for k, v in feeds.items()})]
        d = tmp_path / name
        d.mkdir()
        rec, got = _native(d, plan, blob, feeds, extra=("--runs", "3", "--graph") if name == "toy" else ())
        assert len(got) == len(want) and rec["kernel_calls"] == r.calls, name
        for a, b in zip(got, want):
            assert a.shape == b.shape and np.array_equal(a, b), name
        if name == "toy":
            assert rec["eager_ms"] > 0 and rec["graph_ms"] > 0


@pytest.mark.gpu
def test_native_runner_runs_the_folded_batch_plan_bit_for_bit(ctx, tmp_path):
    """lele_amd.plan/3 (plan.fold_channel_views: channel views, Concat buffers reserved up front, results written into windows of them,
    conv2d_res, copy_view, transpose_cp) is the form bench.py times -- the 7.7 ms Yolo graph.  lele_run executes it natively: the
    Yolo26n-seg look-alike at batch 8, folded, through the Python runner and through lele_run (eagerly, as a recorded graph, and
    scheduled onto lanes as a DAG inside a recorded graph): every output bit for bit, the same number of kernel calls.  A format the
    runner does not know is still refused by name."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from lele_amd.compiler import compile_model
    from lele_amd.lanes import schedule
    from lele_amd.plan import Runner, fold_channel_views, load_weights_bin
    from lele_amd.tensor import TensorView
    from yolo_graph import yolo_onnx
    n = 8
    plan, blob = compile_model(yolo_onnx(n)[0], "yolo26n_seg_shaped_n%d" % n)
    w = load_weights_bin(plan, blob)
    images = np.random.default_rng(8).uniform(0, 1, (n, 3, 640, 640)).astype(np.float32)
    feed = {"images": TensorView(ctx.buf().upload(images))}
    r0 = Runner(plan, w, ctx)
    r0.shapes = {}
    plain = [o.numpy().copy() for o in r0.run(feed)]
    folded = fold_channel_views(plan, r0.shapes)
    assert folded["format"] == "lele_amd.plan/3" and any(st["op"] == "chview" for st in folded["statements"]) \
        and any("window" in st for st in folded["statements"]) and any(st.get("fn") == "conv2d_res" for st in folded["statements"])
    r1 = Runner(folded, w, ctx)
    want = [o.numpy().copy() for o in r1.run(feed)]
    assert all(np.array_equal(a, b) for a, b in zip(plain, want))
    d = tmp_path / "folded"
    d.mkdir()
    rec, got = _native(d, folded, blob, {"images": images}, extra=("--runs", "3", "--graph"))
    assert len(got) == len(want) and rec["kernel_calls"] == r1.calls
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.array_equal(a, b)
    assert rec["graph_ms"] > 0
    # ... and as a DAG (lanes), the schedule bench.py records
    r1.stmt_times = []
    r1.run(feed)
    times = {o: ms for _i, _fn, o, ms in r1.stmt_times}
    dag = schedule(folded, times, lanes=3)
    if dag is not None:
        d2 = tmp_path / "dag"
        d2.mkdir()
        rec2, got2 = _native(d2, dag, blob, {"images": images}, extra=("--runs", "3", "--graph"))
        assert all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got2, want)) and rec2["graph_ms"] > 0
    bad = {"format": "lele_amd.plan/9", "inputs": [], "outputs": [], "slots": [], "weights": {}, "statements": []}
    (tmp_path / "p.json").write_text(json.dumps(bad))
    (tmp_path / "w.bin").write_bytes(b"")
    r = subprocess.run([RUN, str(tmp_path / "p.json"), str(tmp_path / "w.bin"), "--out", str(tmp_path / "o")], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=120)
    assert r.returncode != 0 and "lele_amd.plan/9" in r.stderr and "not supported by the native runner" in r.stderr


@pytest.mark.gpu
def test_native_runner_runs_a_dag_plan(ctx, tmp_path):
    """a plan scheduled onto lanes (lele_amd/lanes.py: "lane" / "wait" / "record" on its statements, a final "join") through lele_run:
    the same bits as the Python runner's sequential plan, eagerly and as a recorded graph with parallel branches"""
    from lele_amd.compiler import compile_model
    from lele_amd.lanes import schedule
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.tensor import TensorView
    from tests.onnx_util import export
    from tests.test_compiler import Vision
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 3, 32, 32, generator=g)
    plan, blob = compile_model(export(Vision().eval(), (x,), opset=13, output_names=("y", "m")), "vision")
    rng = np.random.default_rng(3)
    times = {st["out"][0]: float(rng.uniform(0.001, 0.1)) for st in plan["statements"] if st.get("out")}
    dag = schedule(plan, times, lanes=3, min_gain_ms=-1.0)        # hop lanes wherever two statements are independent
    assert dag is not None and dag["dag"]["lanes"] >= 2 and dag["dag"]["events"] >= 1
    want = [o.numpy() for o in Runner(plan, load_weights_bin(plan, blob), ctx).run({"x": TensorView(ctx.buf().upload(x.numpy()))})]
    rec, got = _native(tmp_path, dag, blob, {"x": x.numpy()}, extra=("--runs", "3", "--graph"))
    assert len(got) == len(want) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(got, want))
    assert rec["graph_ms"] > 0
