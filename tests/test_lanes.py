"""Plans as DAGs (lele_amd/lanes.py + lele_hip_lane_*): independent branches on different HIP streams of one context, recorded as ONE
hipGraph with parallel branches.  The contract: the SAME kernels with the same arguments -- so a DAG plan's outputs equal the sequential
plan's bit for bit, eagerly and replayed, for the measured schedule and for adversarial ones (random per-statement costs and no
hysteresis make the scheduler hop lanes at every opportunity: every cross-lane read-after-write has to carry an event, and every
re-used workspace slot has to be free by happens-before, not by luck of timing)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _toy_dag_plan():
    """x -> a = relu(x); two independent branches b = sigmoid(a), c = exp(a); d = add(b, c)"""
    st = [{"op": "call", "out": ["a"], "fn": "relu", "args": [{"ref": "x"}], "bufs": 1},
          {"op": "call", "out": ["b"], "fn": "sigmoid", "args": [{"ref": "a"}], "bufs": 1},
          {"op": "call", "out": ["c"], "fn": "exp", "args": [{"ref": "a"}], "bufs": 1},
          {"op": "call", "out": ["c2"], "fn": "reshape", "args": [{"ref": "c"}, {"list": [{"int": -1}]}], "bufs": 0},
          {"op": "call", "out": ["b2"], "fn": "reshape", "args": [{"ref": "b"}, {"list": [{"int": -1}]}], "bufs": 0},
          {"op": "call", "out": ["d"], "fn": "add", "args": [{"ref": "b2"}, {"ref": "c2"}], "bufs": 1},
          {"op": "call", "out": ["e"], "fn": "sigmoid", "args": [{"ref": "d"}], "bufs": 1}]
    return {"source": "toy", "format": "lele_amd.plan/2", "inputs": ["x"], "outputs": ["e"], "slots": ["buf_0", "buf_1", "buf_2"], "statements": st, "weights": {}}


def test_scheduler_orders_every_cross_lane_edge_and_frees_slots_by_happens_before():
    """no GPU: the structure of a scheduled plan"""
    from lele_amd.lanes import schedule
    plan = _toy_dag_plan()
    dag = schedule(plan, {"a": 0.01, "b": 0.05, "c": 0.05, "d": 0.01, "e": 0.01}, lanes=2)
    sts = {st["out"][0]: st for st in dag["statements"] if st["out"]}
    assert dag["dag"]["lanes"] == 2 and sts["b"]["lane"] != sts["c"]["lane"]          # the two branches overlap
    side = sts["b"] if sts["b"]["lane"] != sts["a"]["lane"] else sts["c"]
    start = dag["statements"][0]
    assert start["op"] == "join" and start["wait"] == [] and "record" in start        # the run's starting point on lane 0
    assert side["wait"] == [start["record"], sts["a"]["record"]]                       # fork: waits for the start and for a
    assert sts["d"]["lane"] in (0, 1) and set(sts["d"].get("wait", [])) == {side["record"]} if sts["d"]["lane"] != side["lane"] else True
    assert dag["statements"][-1]["op"] == "join"
    # b and c are alive together: they can never share a slot; e may not take the slot of a value its lane has not seen die
    assert sts["b"]["slots"] != sts["c"]["slots"] and sts["a"]["slots"] != sts["b"]["slots"] and sts["a"]["slots"] != sts["c"]["slots"]
    assert dag["dag"]["modelled_makespan_ms"] < dag["dag"]["modelled_sequential_ms"]
    # constructs the pass does not order are refused, not guessed at
    assert schedule(dict(plan, statements=plan["statements"] + [{"op": "if", "out": []}])) is None
    one = schedule(plan, {}, lanes=1)
    assert one["dag"]["lanes"] == 1 and all(st.get("lane", 0) == 0 for st in one["statements"])


def _random_plan(rng, n):
    """a random SSA plan: unary / binary statements over earlier values, views of earlier values, a two-result split now and then"""
    st, vals = [], ["x", "y"]     # two inputs: statements that read only inputs have no producer inside the run
    for i in range(n):
        kind = rng.choice(["unary", "binary", "view", "split"], p=[0.35, 0.4, 0.15, 0.1])
        a = vals[int(rng.integers(max(0, len(vals) - 6), len(vals)))]
        if kind == "unary":
            st.append({"op": "call", "out": ["v%d" % i], "fn": "relu", "args": [{"ref": a}], "bufs": 1})
        elif kind == "binary":
            b = vals[int(rng.integers(0, len(vals)))]
            st.append({"op": "call", "out": ["v%d" % i], "fn": "add", "args": [{"ref": a}, {"ref": b}], "bufs": 1})
        elif kind == "view":
            st.append({"op": "call", "out": ["v%d" % i], "fn": "reshape", "args": [{"ref": a}, {"list": [{"int": -1}]}], "bufs": 0})
        else:
            st.append({"op": "call", "out": ["v%d" % i, "w%d" % i], "fn": "split", "args": [{"ref": a}, {"int": 0}, {"list": [{"int": 1}, {"int": 1}]}], "bufs": 2})
            vals.append("w%d" % i)
        vals.append("v%d" % i)
    outs = [vals[-1], vals[len(vals) // 2]]
    return {"source": "random", "format": "lele_amd.plan/2", "inputs": ["x", "y"], "outputs": outs, "slots": [], "statements": st, "weights": {}}


def test_random_plans_every_hazard_is_ordered_by_happens_before():
    """An INDEPENDENT check of the scheduler (no GPU): rebuild happens-before from what the emitted plan says -- program order on each lane
    plus record -> wait edges, transitively closed -- and verify (a) every read of a value is ordered after its producer, (b) two
    values share a workspace slot only if EVERY access of the earlier one (its writer and all readers, through views) happens-before
    the writer of the later one, (c) the final join waits for the tail of every side lane, (d) outputs never lose their slot,
    (e) EVERY statement of a side lane -- also one that reads only plan inputs -- is ordered after the run's starting point on lane 0,
    so that with the final join of the previous run nothing of run k + 1 can overtake anything of run k.
    200 random plans, random costs, 2-4 lanes, with and without hysteresis."""
    from lele_amd.lanes import schedule
    rng = np.random.default_rng(2026)
    checked_pairs = forks = 0
    for trial in range(200):
        plan = _random_plan(rng, int(rng.integers(8, 45)))
        times = {st["out"][0]: float(rng.uniform(0.001, 0.1)) for st in plan["statements"]}
        dag = schedule(plan, times, lanes=int(rng.integers(2, 5)), min_gain_ms=float(rng.choice([-1.0, 0.004, 0.03])))
        sts = dag["statements"]
        for st in sts:
            if st["op"] == "join":
                st["lane"] = 0          # the start / the final join are points on lane 0
        dev = [i for i, st in enumerate(sts) if "lane" in st]
        n = len(sts)
        hb = np.zeros((n, n), bool)
        last_on_lane, recorder = {}, {}
        for i in dev:
            st = sts[i]
            if st["lane"] in last_on_lane:
                hb[last_on_lane[st["lane"]], i] = True
            for e in st.get("wait", []):
                hb[recorder[e], i] = True
            last_on_lane[st["lane"]] = i
            if "record" in st:
                recorder[st["record"]] = i
        for k in dev:                                   # transitive closure (plan order is a topological order)
            hb[:, k] |= (hb[:, dev] & hb[dev, k][None, :]).any(axis=1)
        # value -> root buffer, producer of every value, accesses of every root
        if any(sts[i]["lane"] != 0 for i in dev):                                                                                     # (e)
            assert sts[0]["op"] == "join" and "record" in sts[0]
            for i in dev[1:]:
                assert hb[0, i], "trial %d: statement %d on lane %d is not ordered after the start of the run" % (trial, i, sts[i]["lane"])
        dev = [i for i in dev if sts[i]["op"] != "join"]
        root, producer = {"x": "x", "y": "y"}, {}
        for i, st in enumerate(sts):
            if st["op"] != "call":
                continue
            for o in st["out"]:
                producer[o] = i
                root[o] = root[st["args"][0]["ref"]] if st.get("bufs", 1) == 0 else o
        writer = {}
        acc = {}
        for i in dev:
            st = sts[i]
            reads = [a["ref"] for a in st["args"] if isinstance(a, dict) and "ref" in a]
            for r in reads:
                rt = root[r]
                acc.setdefault(rt, set()).add(i)
                w = writer.get(rt)
                assert w is None or hb[w, i], "trial %d: statement %d reads %s before its producer %d is ordered" % (trial, i, r, w)   # (a)
            for o in st["out"]:
                writer[o] = i
                acc.setdefault(o, set()).add(i)
        slot_of = {}
        for i in dev:
            for o, s_ in zip(sts[i]["out"], sts[i].get("slots", [])):
                slot_of[o] = s_
        pinned = {root[o] for o in dag["outputs"]}
        by_slot = {}
        for o, s_ in slot_of.items():
            by_slot.setdefault(s_, []).append(o)
        for s_, names in by_slot.items():
            names.sort(key=lambda o: writer[o])
            for u, w in zip(names, names[1:]):
                assert u not in pinned, "trial %d: output %s lost its slot to %s" % (trial, u, w)                                   # (d)
                for a in acc[u]:
                    assert hb[a, writer[w]], "trial %d: %s takes the slot of %s while statement %d may still touch it" % (trial, w, u, a)   # (b)
                    checked_pairs += 1
        join = sts[-1]
        assert join["op"] == "join"
        for lane, last in last_on_lane.items():
            last = max(i for i in dev if sts[i]["lane"] == lane) if any(sts[i]["lane"] == lane for i in dev) else last
            if lane != 0:
                assert sts[last].get("record") in join["wait"], "trial %d: lane %d is not joined" % (trial, lane)                     # (c)
        forks += dag["dag"]["events"]
    assert checked_pairs > 500 and forks > 500      # the property was exercised, not vacuously true


def _graph_outputs(ctx, runner, feed, replays=3):
    ctx.sync()
    ctx.graph_begin()
    res = runner.run(feed)
    g = ctx.graph_end()
    for _ in range(replays):
        g.launch()
    ctx.sync()
    out = [o.numpy().copy() for o in res]
    g.close()
    return out


@pytest.mark.gpu
def test_lanes_through_the_c_abi(ctx):
    """two lanes by hand: b on lane 1 after an event of lane 0, joined back; eager and recorded"""
    from lele_amd import kernels as K
    x = np.random.default_rng(0).standard_normal((64, 1024)).astype(np.float32)
    xd = ctx.buf().upload(x)
    bufs = [ctx.buf() for _ in range(4)]
    base = ctx.lane_events(2)

    def run():
        a = K.relu(xd, out=bufs[0], ctx=ctx)
        ctx.lane_record(base)
        ctx.lane_set(1)
        ctx.lane_wait(base)
        b = K.sigmoid(a, out=bufs[1], ctx=ctx)
        ctx.lane_record(base + 1)
        ctx.lane_set(0)
        c = K.exp(a, out=bufs[2], ctx=ctx)
        ctx.lane_wait(base + 1)
        return K.add(b, c, out=bufs[3], ctx=ctx)
    want = K.add(K.sigmoid(K.relu(x, ctx=ctx), ctx=ctx), K.exp(K.relu(x, ctx=ctx), ctx=ctx), ctx=ctx).numpy()
    got = run().numpy()
    assert np.array_equal(got, want)
    ctx.sync()
    ctx.graph_begin()
    res = run()
    g = ctx.graph_end()
    for _ in range(3):
        g.launch()
    ctx.sync()
    assert np.array_equal(res.numpy(), want)
    g.close()
    from lele_amd import _lib
    ctx.lane_set(1)
    with pytest.raises(_lib.LeleError, match="lane 1 is current"):
        ctx.graph_begin()
    ctx.lane_set(0)


@pytest.mark.gpu
def test_dag_plan_with_two_inputs_back_to_back_eager_runs(ctx):
    """a statement that reads only plan inputs has no producer inside the run.  Two inputs, the second consumed FIRST on a
    side lane; the inputs are produced by kernels on lane 0's stream right before the run (no sync), and the plan runs 6 times back
    to back eagerly without a sync in between, each time on fresh inputs -- every run's outputs must be that run's, and event ids
    handed back by a closed Runner are reused."""
    from lele_amd import kernels as K
    from lele_amd.lanes import schedule
    from lele_amd.plan import Runner
    st = [{"op": "call", "out": ["a"], "fn": "exp", "args": [{"ref": "x"}], "bufs": 1},
          {"op": "call", "out": ["b"], "fn": "sigmoid", "args": [{"ref": "y"}], "bufs": 1},
          {"op": "call", "out": ["b1"], "fn": "tanh", "args": [{"ref": "b"}], "bufs": 1},
          {"op": "call", "out": ["a1"], "fn": "relu", "args": [{"ref": "a"}], "bufs": 1},
          {"op": "call", "out": ["c"], "fn": "add", "args": [{"ref": "a1"}, {"ref": "b1"}], "bufs": 1},
          {"op": "call", "out": ["d"], "fn": "sigmoid", "args": [{"ref": "c"}], "bufs": 1}]
    plan = {"source": "toy2", "format": "lele_amd.plan/2", "inputs": ["x", "y"], "outputs": ["d"], "slots": ["buf_0", "buf_1", "buf_2"],
            "statements": st, "weights": {}}
    dag = schedule(plan, {"a": 0.05, "b": 0.05, "b1": 0.05, "a1": 0.05, "c": 0.01, "d": 0.01}, lanes=2, min_gain_ms=-1.0)
    lanes_used = {s_["lane"] for s_ in dag["statements"] if "lane" in s_}
    assert lanes_used == {0, 1}
    side_first = next(s_ for s_ in dag["statements"] if s_.get("lane") == 1)
    assert dag["statements"][0]["record"] in side_first["wait"]
    before = ctx._next_event
    r = Runner(dag, {}, ctx)
    rng = np.random.default_rng(4)
    base = [rng.standard_normal((1 << 20,)).astype(np.float32) for _ in range(2)]
    xs, ys = ctx.buf().upload(base[0]), ctx.buf().upload(base[1])
    xin, yin = ctx.buf(), ctx.buf()
    outs, keep = [], [ctx.buf() for _ in range(6)]
    for k in range(6):     # inputs produced on lane 0's stream by kernels (a big reduction chain makes them late), then the plan at once
        sc = ctx.buf().upload(np.array([1.0 + k], np.float32))
        xk = K.mul(xs, sc, out=xin, ctx=ctx)
        yk = K.mul(ys, sc, out=yin, ctx=ctx)
        d = r.run({"x": xk, "y": yk})[0]
        outs.append(K.view_copy(d, [], out=keep[k], ctx=ctx))
    ctx.sync()
    for k in range(6):
        x, y = base[0] * np.float32(1.0 + k), base[1] * np.float32(1.0 + k)
        want = K.sigmoid(K.add(K.relu(K.exp(x, ctx=ctx), ctx=ctx), K.tanh(K.sigmoid(y, ctx=ctx), ctx=ctx), ctx=ctx), ctx=ctx).numpy()
        assert np.array_equal(outs[k].numpy(), want), "run %d" % k
    r.close()
    r2 = Runner(dag, {}, ctx)
    assert r2.event_base == r.event_base and ctx._next_event == before + dag["dag"]["events"]     # the id range came back
    r2.close()


def _check_dag(ctx, plan, weights, feed, times=None, **kw):
    from lele_amd.lanes import schedule
    from lele_amd.plan import Runner
    seq = Runner(plan, weights, ctx)
    want = [o.numpy().copy() for o in seq.run(feed)]
    dag_plan = schedule(plan, times, **kw)
    assert dag_plan is not None
    r = Runner(dag_plan, weights, ctx)
    got = [o.numpy().copy() for o in r.run(feed)]
    assert all(np.array_equal(a, b) for a, b in zip(got, want)), "eager DAG run differs from the sequential plan"
    for k in range(2):
        again = _graph_outputs(ctx, r, feed)
        assert all(np.array_equal(a, b) for a, b in zip(again, want)), "recorded DAG differs from the sequential plan (round %d)" % k
    return dag_plan


@pytest.mark.gpu
def test_dag_plans_equal_sequential_plans_bit_for_bit(ctx):
    from yolo_graph import yolo_onnx
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, fold_channel_views, load_weights_bin
    from lele_amd.tensor import TensorView
    nb = 4
    plan, blob = compile_model(yolo_onnx(nb)[0], "yolo_n%d" % nb)
    w = load_weights_bin(plan, blob)
    images = np.random.default_rng(5).uniform(0, 1, (nb, 3, 640, 640)).astype(np.float32)
    feed = {"images": TensorView(ctx.buf().upload(images))}
    r0 = Runner(plan, w, ctx)
    r0.shapes = {}
    r0.stmt_times = []
    r0.run(feed)
    r0.stmt_times = []
    r0.run(feed)
    times = {o: ms for _i, _fn, o, ms in r0.stmt_times}
    r0.stmt_times = None
    dag = _check_dag(ctx, plan, w, feed, times, lanes=3)
    assert dag["dag"]["lanes"] >= 2 and dag["dag"]["modelled_makespan_ms"] < dag["dag"]["modelled_sequential_ms"]
    # adversarial schedules: random costs, no hysteresis -> lane hops wherever two statements are independent
    rng = np.random.default_rng(9)
    for trial in range(3):
        rnd = {k: float(rng.uniform(0.001, 0.2)) for k in times}
        d = _check_dag(ctx, plan, w, feed, rnd, lanes=4, min_gain_ms=-1.0)
        assert d["dag"]["events"] > dag["dag"]["events"]
    # the batch form (channel views, windows written in place): format 3
    folded = fold_channel_views(plan, r0.shapes)
    d3 = _check_dag(ctx, folded, w, feed, times, lanes=3)
    assert d3["dag"]["lanes"] >= 2
    for trial in range(2):
        rnd = {k: float(rng.uniform(0.001, 0.2)) for k in times}
        _check_dag(ctx, folded, w, feed, rnd, lanes=4, min_gain_ms=-1.0)


@pytest.mark.gpu
def test_sensevoice_shaped_dag_plan(ctx):
    """a 3-layer SenseVoice-shaped encoder: the FSMN memory block of a layer runs beside its attention"""
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, load_weights_bin
    from lele_amd.tensor import TensorView
    from sensevoice_graph import Encoder, encoder_onnx
    enc = Encoder(ctx, 3)
    for b, t in ((4, 171), (1, 504)):
        # (without round 6's half-layer statements: with them the memory block is inside the projection's statement and a layer is a chain)
        plan, blob = compile_model(encoder_onnx(enc, b), "sv", half_layer_folds=False)
        w = load_weights_bin(plan, blob)
        x = np.random.default_rng(t).standard_normal((b, t, 560)).astype(np.float32)
        feed = {"feats": TensorView(ctx.buf().upload(x))}
        r0 = Runner(plan, w, ctx)
        r0.run(feed)
        r0.stmt_times = []
        r0.run(feed)
        times = {o: ms for _i, _fn, o, ms in r0.stmt_times}
        r0.stmt_times = None
        dag = _check_dag(ctx, plan, w, feed, times, lanes=2)
        fsmn = [st for st in dag["statements"] if st.get("fn") == "depthwise_conv1d_tlc"]
        attn = [st for st in dag["statements"] if st.get("fn") == "attention_view"]
        assert fsmn and attn and any(f["lane"] != a["lane"] for f, a in zip(fsmn, attn)), dag["dag"]
        rng = np.random.default_rng(1)
        _check_dag(ctx, plan, w, feed, {k: float(rng.uniform(0.001, 0.1)) for k in times}, lanes=4, min_gain_ms=-1.0)
