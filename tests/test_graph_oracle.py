"""BASELINE configs[4] at the level the reference works at -- the GRAPH -- against the CPU oracle (VERDICT r4 "Next" item 2).

oracle/plan_ref.py executes a plan statement by statement over the oracle's restatements of lele's kernels (im2col convolution with the
AVX2 epilogues, k-ordered GEMM, polynomial softmax / sigmoid / SiLU, stable top-k, ...).  Here the device's batch-64 forward of
  (a) the Yolo26n-seg-shaped network of tools/yolo_graph.py (compiled from ONNX by lele_amd.compiler), and
  (b) the reference's OWN generated Yolo26n-seg call sequence (lifted from examples/yolo26n-seg/src/yolo26seg.rs, re-batched)
is held against the oracle's forward of the same image, for images 0 and 63 of the batch:
  * the prototype map, and the three pre-top-k tensors the detection tail selects from (boxes [A, 4], class scores [A, 80], mask
    coefficients [A, 32]) plus the per-anchor best score the first top-k ranks: element for element at the north_star bar
    (tests/parity.py close_f32, 1e-4);
  * the detections [300, 38]: lele's top-k is a STABLE descending sort (src/kernels/conv2d.rs:1385-1435), so the oracle's order is the
    reference's order.  Two f32 implementations may rank two candidates differently only where the oracle's own scores are within the
    tolerance of each other; everywhere else the device row must BE the oracle's row.  Checked per row r of the device output:
      - identity: its (box, coefficients) are those of exactly one anchor a of the oracle's tensors and its class c is an integer
        in [0, 80); no (a, c) occurs twice;
      - content: box, score, class, coefficients equal the oracle's values for (a, c) at 1e-4;
      - rank: the oracle's score of (a, c) is within the tolerance of the oracle's r-th best score (the device's order is a valid
        descending order of the oracle's scores up to the tolerance), and where the oracle's r-th score is separated from both
        neighbours by more than twice the tolerance the device row is the oracle's row r itself.
    The number of rows inside such near-ties is printed and bounded.
The synthetic weights are calibrated on the oracle (plan_ref.calibrate: every convolution's pre-activation has unit variance on image
0), so that all 118 layers carry signal -- plain N(0, 1/sqrt(fan_in)) weights let the scores collapse to 0.515 +- 1e-5."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from parity import close_f32  # noqa: E402

TOL = 1e-4
LIFTED = os.path.join(ROOT, "_lifted", "yolo26seg_plan.json")


# ------------------------------------------------------------------------------------------------------------- CPU: the executor itself
def _toy_plan():
    w = {0: ["weight_f32", 432, [4, 3, 3, 3]], 432: ["weight_f32", 16, [4]], 448: ["weight_f32", 16, [4]], 464: ["weight_i64", 8, [1]]}
    W = lambda o: [w[o][0], o, w[o][1], w[o][2]]   # noqa: E731
    L = lambda *v: {"list": [{"int": int(i)} for i in v]}   # noqa: E731
    st = [
        {"op": "call", "out": ["a"], "fn": "conv2d_silu", "args": [{"ref": "images"}, {"weight": W(0)}, {"some": {"weight": W(432)}}, L(1, 1), {"int": 1}, L(1, 1, 1, 1), L(2, 2), {"slot": "buf_0"}]},
        {"op": "ints", "out": ["sizes"], "value": [2, 2]},
        {"op": "call", "out": ["parts"], "fn": "split_owned", "args": [{"ref": "a"}, {"int": 1}, {"ref": "sizes"}]},
        {"op": "swap_remove", "out": ["hi"], "list": "parts", "index": 1},
        {"op": "swap_remove", "out": ["lo"], "list": "parts", "index": 0},
        {"op": "call", "out": ["r"], "fn": "resize_nearest", "args": [{"ref": "hi"}, {"some": {"weight_list": W(448)}}, {"none": True}, {"str": "asymmetric"}, {"slot": "buf_1"}]},
        {"op": "call", "out": ["s"], "fn": "sigmoid", "args": [{"ref": "r"}, {"slot": "buf_0"}]},
        {"op": "call", "out": ["v"], "fn": "reshape", "args": [{"ref": "s"}, L(1, 2, -1)]},
        {"op": "newbuf", "out": ["bv"]}, {"op": "newbuf", "out": ["bi"]},
        {"op": "call", "out": ["tv", "ti"], "fn": "topk", "args": [{"ref": "v"}, {"weight_scalar": W(464)}, {"int": -1}, {"bool": True}, {"bool": True}, {"buf": "bv"}, {"buf": "bi"}]},
        {"op": "alias", "out": ["c"], "src": "tv"},
        {"op": "call", "out": ["output0"], "fn": "concat", "args": [{"refs": ["c", "ti"]}, {"int": -1}, {"slot": "buf_0"}]},
    ]
    return {"source": "toy", "inputs": ["images"], "outputs": ["output0"], "slots": ["buf_0", "buf_1"], "statements": st, "weights": {str(k): v for k, v in w.items()}}


def test_plan_ref_runs_a_lifted_style_plan_as_the_oracle_composition(orc):
    """the executor adds nothing of its own: a lifted-style plan through it == the same oracle / numpy calls written out by hand"""
    from oracle import npref, plan_ref
    plan = _toy_plan()
    rng = np.random.default_rng(3)
    W = {0: rng.standard_normal((4, 3, 3, 3)).astype(np.float32), 432: rng.standard_normal(4).astype(np.float32),
         448: np.array([1, 1, 2, 2], np.float32), 464: np.array([5], np.int64)}
    x = rng.standard_normal((1, 3, 12, 12)).astype(np.float32)
    taps = {"r": None}
    got, = plan_ref.run(plan, W, {"images": x}, taps=taps)
    a = orc.conv2d_im2col(x, W[0], W[432], [1, 1], 1, [1, 1, 1, 1], [2, 2], "silu")
    r = npref.resize_nearest(a[:, 2:], 12, 12, True)
    tv, ti = npref.topk(orc.unary("sigmoid", r).reshape(1, 2, -1), 5, True)
    assert np.array_equal(got, np.concatenate([tv, ti], -1)) and np.array_equal(taps["r"], r) and got.shape == (1, 2, 10)
    with pytest.raises(NotImplementedError):
        plan_ref.PlanRef(plan, W).call("no_such_kernel", [])


def test_calibrate_gives_every_convolution_unit_variance(orc):
    from oracle import plan_ref
    plan = _toy_plan()
    rng = np.random.default_rng(4)
    W = {0: (rng.standard_normal((4, 3, 3, 3)) * 7).astype(np.float32), 432: rng.standard_normal(4).astype(np.float32),
         448: np.array([1, 1, 2, 2], np.float32), 464: np.array([5], np.int64)}
    x = rng.standard_normal((1, 3, 12, 12)).astype(np.float32)
    Wc = plan_ref.calibrate(plan, W, {"images": x})
    pre = orc.conv2d_im2col(x, Wc[0], Wc[432], [1, 1], 1, [1, 1, 1, 1], [2, 2], None)
    assert abs(float(pre.std()) - 1.0) < 1e-3 and np.array_equal(Wc[448], W[448]) and Wc[464].dtype == np.int64
    assert np.allclose((Wc[0] / W[0]).reshape(-1), (Wc[432] / W[432])[0], rtol=1e-5)          # weight and bias by the same factor: the layer's function up to scale


# ------------------------------------------------------------------------------------------------------------- the detection tail, row by row
def head_taps(plan):
    """names of the tensors a Yolo detection tail selects from, by structure: the sources of the plan's GatherElements statements
    (boxes / class scores / coefficients, [N, A, c]) and the input of its first TopK (best class score per anchor, [N, A])"""
    calls = [s for s in plan["statements"] if s["op"] == "call"]
    srcs = [s["args"][0]["ref"] for s in calls if s["fn"] == "gather_elements"]
    first = next(s["args"][0]["ref"] for s in calls if s["fn"] == "topk")
    return srcs, first


def soften_class_logits(plan, weights, key, factor=0.25):
    """the convolution that produces the class LOGITS of each level (80 output channels, no activation behind it): weight
    and bias times `factor`.  With unit-variance logits the best 300 of 672 000 sigmoid scores all sit above 0.99, a few 1e-5 apart --
    every rank inside a near-tie; at a quarter of that the leading ranks are separated by more than the tolerance and are checked row
    for row against the oracle's stable order.  Returns a new weights dict."""
    activated = set()          # values an activation reads: sigmoid(x) / silu(x) (the lifted plan keeps Conv + Sigmoid + Mul apart)
    for st in plan["statements"]:
        if st.get("fn") in ("sigmoid", "silu") and isinstance(st["args"][0], dict):
            activated.add(st["args"][0].get("ref"))
    out = dict(weights)
    n = 0
    for st in plan["statements"]:
        if st.get("fn") == "conv2d" and st["args"][1]["weight"][3][0] == 80 and st["out"][0] not in activated:
            bias = st["args"][2]["some"] if "some" in st["args"][2] else st["args"][2]     # lifted: Some(&weight); compiled: the weight itself
            for node in (st["args"][1]["weight"], bias["weight"]):
                out[key(node)] = (np.asarray(weights[key(node)]) * np.float32(factor)).astype(np.float32)
            n += 1
    assert n == 3, n
    return out


def check_detections(dev, orc_det, box, cls, coef, what):
    """dev, orc_det [300, 38]; box [A, 4], cls [A, 80], coef [A, 32] = the ORACLE's pre-top-k tensors.  See the module docstring."""
    k, width = dev.shape
    nc = cls.shape[1]
    assert width == 4 + 2 + coef.shape[1] and orc_det.shape == dev.shape
    sig = np.concatenate([box, coef], axis=1).astype(np.float64)                 # an anchor's signature: 36 numbers
    rms = float(np.sqrt(np.mean(np.square(sig))))
    seen, rows_tied, rows_exact = set(), 0, 0
    o_scores = orc_det[:, 4].astype(np.float64)
    # the candidate right below the cut: the best score the oracle did NOT select (ties at rank k)
    flat = np.sort(cls.reshape(-1).astype(np.float64))[::-1]
    below = flat[k] if flat.size > k else -np.inf
    for r in range(k):
        row = dev[r].astype(np.float64)
        c = row[5]
        assert c == np.floor(c) and 0 <= c < nc, "%s row %d: class %r" % (what, r, c)
        c = int(c)
        d = np.abs(sig - np.concatenate([row[:4], row[6:]])[None, :])
        ok = (d <= TOL * np.abs(sig) + TOL * rms + 1e-7).all(axis=1)
        cand = np.nonzero(ok)[0]
        assert cand.size >= 1, "%s row %d: no anchor of the oracle has this box / coefficient row" % (what, r)
        # duplicates of a signature are equal content; the score picks among them
        a = int(cand[np.argmin(np.abs(cls[cand, c].astype(np.float64) - row[4]))])
        assert (a, c) not in seen, "%s row %d: (anchor %d, class %d) selected twice" % (what, r, a, c)
        seen.add((a, c))
        s_ref = float(cls[a, c])
        bar = TOL * abs(s_ref) + 1e-7
        assert abs(row[4] - s_ref) <= bar, "%s row %d: score %.9g, oracle %.9g" % (what, r, row[4], s_ref)
        assert abs(s_ref - o_scores[r]) <= 2 * bar, "%s row %d: oracle score %.9g of the device's pick is not the oracle's rank-%d score %.9g" % (what, r, s_ref, r, o_scores[r])
        hi = o_scores[r - 1] if r else np.inf
        lo = o_scores[r + 1] if r + 1 < k else below
        if min(hi - o_scores[r], o_scores[r] - lo) > 4 * bar:      # no candidate within the tolerance of this rank: the row is decided
            close_f32(dev[r], orc_det[r], TOL, "%s row %d (decided rank)" % (what, r))
            rows_exact += 1
        else:
            rows_tied += 1
    return rows_exact, rows_tied


def _compare_image(dev_outs, dev_taps, i, ref_outs, ref_taps, srcs, first, what):
    det = next(o for o in dev_outs if o.ndim == 3)
    proto = next(o for o in dev_outs if o.ndim == 4)
    rdet = next(o for o in ref_outs if o.ndim == 3)
    rproto = next(o for o in ref_outs if o.ndim == 4)
    close_f32(proto[i:i + 1], rproto, TOL, "%s: prototype map of image %d" % (what, i))
    for name in srcs + [first]:
        close_f32(dev_taps[name][i:i + 1], ref_taps[name], TOL, "%s: %s of image %d" % (what, name, i))
    by_width = {ref_taps[n].shape[-1]: ref_taps[n][0] for n in srcs}
    widths = sorted(by_width)
    box, coef, cls = by_width[widths[0]], by_width[widths[1]], by_width[widths[2]]
    assert box.shape[1] == 4 and cls.shape[1] > coef.shape[1]
    return check_detections(det[i], rdet[0], box, cls, coef, "%s image %d" % (what, i))


@pytest.mark.gpu
def test_c5_lookalike_graph_batch_64_against_the_oracle_forward(ctx):
    from yolo_graph import yolo_onnx
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, fold_channel_views, load_weights_bin
    from lele_amd.tensor import TensorView
    from oracle import plan_ref
    n = 64
    p1, b1 = compile_model(yolo_onnx(1)[0], "yolo26n_seg_shaped_n1")
    pn, bn = compile_model(yolo_onnx(n)[0], "yolo26n_seg_shaped_n%d" % n)
    from lele_amd.plan import weight_key
    rng = np.random.default_rng(64)
    images = rng.uniform(0, 1, (n, 3, 640, 640)).astype(np.float32)
    w1, wn = load_weights_bin(p1, b1), load_weights_bin(pn, bn)
    real = {k for k, v in w1.items() if v.dtype == np.float32}       # the two exports differ in their integer shape constants only
    assert real == {k for k, v in wn.items() if v.dtype == np.float32} and all(np.array_equal(w1[k], wn[k]) for k in real)
    w1 = soften_class_logits(p1, plan_ref.calibrate(p1, w1, {"images": images[:1]}), weight_key)
    wn = {k: (w1[k] if k in real else v) for k, v in wn.items()}
    srcs, first = head_taps(p1)
    assert (srcs, first) == head_taps(pn) and len(srcs) == 3
    big = Runner(pn, wn, ctx)
    big.taps = {k: None for k in srcs + [first]}
    big.shapes = {}
    feed = {"images": TensorView(ctx.buf().upload(images))}
    outs = [o.numpy().copy() for o in big.run(feed)]
    taps, big.taps = big.taps, None
    folded = Runner(fold_channel_views(pn, big.shapes), wn, ctx)            # what bench.py times: the same bits
    assert all(np.array_equal(a, o.numpy()) for a, o in zip(outs, folded.run(feed)))
    tied = 0
    for i in (0, n - 1):
        rt = {k: None for k in srcs + [first]}
        ref = plan_ref.run(p1, w1, {"images": images[i:i + 1]}, taps=rt)
        exact, t = _compare_image(outs, taps, i, ref, rt, srcs, first, "look-alike")
        print("look-alike image %d: %d rows decided and equal row for row, %d rows inside near-ties of the oracle's own scores" % (i, exact, t))
        tied += t
        _rows_record("look-alike image %d" % i, exact, t)
    assert tied <= MAX_TIED_LOOKALIKE     # every tied row passed the identity / content / rank checks, the others ARE the oracle's rows


# Rows of the 300 detections per image that sit inside near-ties of the ORACLE'S OWN scores (so that "the oracle's row at this rank" is
# not defined to the tolerance), summed over the two images compared.  The bounds are what the builder's runs printed
# (profiles/r06_graph_oracle_rows.json) plus a margin of 20 rows: a regression that blurs more ranks fails here.
MAX_TIED_LOOKALIKE, MAX_TIED_GENERATED = 247 + 262 + 20, 244 + 255 + 20


def _rows_record(what, decided, tied):
    path = os.path.join(ROOT, "gpurun_out", "graph_oracle_rows.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rec = json.load(open(path)) if os.path.exists(path) else {}
    rec[what] = {"rows_decided_and_equal": int(decided), "rows_inside_near_ties": int(tied)}
    json.dump(rec, open(path, "w"), indent=1)


@pytest.mark.gpu
def test_c5_reference_generated_graph_batch_64_against_the_oracle_forward(ctx):
    """the reference's own call sequence (untracked lifted artefact: tools/lift_generated.py lift; skipped by name when absent)"""
    if not os.path.exists(LIFTED):
        pytest.skip("no lifted Yolo26n-seg plan in this checkout (tools/lift_generated.py lift <reference>/examples/yolo26n-seg/src/yolo26seg.rs)")
    import lift_generated as L
    from lele_amd.plan import Runner, fold_channel_views, fuse_sigmoid_mul, rebatch_lifted, replan_lifted
    from lele_amd.tensor import TensorView
    from oracle import plan_ref
    n = 64
    plan = json.load(open(LIFTED))
    name = plan["inputs"][-1]
    rng = np.random.default_rng(64)
    images = rng.uniform(0, 1, (n, 3, 640, 640)).astype(np.float32)
    raw = soften_class_logits(plan, plan_ref.calibrate(plan, L.synth_weights(plan, dict(L.DEFAULT_CONSTS)), {name: images[:1]}), lambda node: node[1])
    srcs, first = head_taps(plan)
    assert len(srcs) == 3
    r1 = Runner(plan, raw, ctx)
    r1.shapes = {}
    r1.run({name: TensorView(ctx.buf().upload(images[:1]))})
    p2 = replan_lifted(fuse_sigmoid_mul(plan, r1.shapes), r1.shapes)
    w2 = {k: raw[int(k.split(":")[0])] for k in p2["weights"]}
    pn = rebatch_lifted(p2, n, r1.shapes)
    big = Runner(pn, w2, ctx)
    big.taps = {k: None for k in srcs + [first]}
    big.shapes = {}
    feed = {name: TensorView(ctx.buf().upload(images))}
    outs = [o.numpy().copy() for o in big.run(feed)]
    taps, big.taps = big.taps, None
    assert all(v is not None for v in taps.values()), [k for k, v in taps.items() if v is None]
    folded = Runner(fold_channel_views(pn, big.shapes), w2, ctx)
    assert all(np.array_equal(a, o.numpy()) for a, o in zip(outs, folded.run(feed)))
    tied = 0
    for i in (0, n - 1):
        rt = {k: None for k in srcs + [first]}
        ref = plan_ref.run(plan, raw, {name: images[i:i + 1]}, taps=rt)
        exact, t = _compare_image(outs, taps, i, ref, rt, srcs, first, "generated graph")
        print("generated graph image %d: %d rows decided and equal row for row, %d rows inside near-ties of the oracle's own scores" % (i, exact, t))
        tied += t
        _rows_record("generated graph image %d" % i, exact, t)
    assert tied <= MAX_TIED_GENERATED


# ------------------------------------------------------------------------------------------------ configs[2] / configs[3]: the whole recogniser
def logits_agreement(dev, ref):
    """The figures examples/sensevoice/tests/e2e_test.rs:126-189 judges the model by -- mean absolute logit difference and arg-max
    agreement per frame -- plus what makes the second one decidable: at a frame whose arg-max ids differ, how far the DEVICE's pick
    sits below the ORACLE's maximum IN THE ORACLE'S OWN LOGITS (a near-tie of the oracle, or a wrong answer?)."""
    dev, ref = np.asarray(dev, np.float64), np.asarray(ref, np.float64)
    rms = float(np.sqrt(np.mean(ref * ref)))
    a_dev, a_ref = dev.argmax(-1), ref.argmax(-1)
    differ = a_dev != a_ref
    gap = (np.take_along_axis(ref, a_ref[..., None], -1) - np.take_along_axis(ref, a_dev[..., None], -1))[..., 0]
    return {"frames": int(a_ref.size), "argmax_agreement": float(1.0 - differ.mean()), "mae": float(np.abs(dev - ref).mean()),
            "mae_over_rms": float(np.abs(dev - ref).mean() / rms), "max_abs_over_rms": float(np.abs(dev - ref).max() / rms),
            "worst_gap_at_a_differing_frame_over_rms": float(gap[differ].max() / rms) if differ.any() else 0.0, "logits_rms": rms}


def _sensevoice_logits(ctx, enc, batch, feats, exact):
    from lele_amd.compiler import compile_model
    from lele_amd.plan import Runner, load_weights_bin
    from sensevoice_graph import encoder_onnx
    plan, blob = compile_model(encoder_onnx(enc, batch), "sensevoice_shaped")
    w = load_weights_bin(plan, blob)
    old = os.environ.get("LELE_HIP_ATTENTION_EXACT")
    try:
        out = {}
        for name, flag in (("shipped", None),) + ((("exact", "1"),) if exact else ()):
            os.environ.pop("LELE_HIP_ATTENTION_EXACT", None)
            if flag:
                os.environ["LELE_HIP_ATTENTION_EXACT"] = flag
            out[name] = Runner(plan, w, ctx).run({"feats": feats})[0].numpy().copy()
    finally:
        os.environ.pop("LELE_HIP_ATTENTION_EXACT", None)
        if old is not None:
            os.environ["LELE_HIP_ATTENTION_EXACT"] = old
    return plan, w, out


# What the builder's runs printed (profiles/r06_sensevoice_graph_oracle.json: agreement 0.91-0.92, MAE 0.29-0.33 = 3.4-3.9 % of the
# logits' rms for BOTH attention paths and for the two device paths against each other), with margin.  SV_MAX_MAE is the reference's own
# bar for this model (examples/sensevoice/tests/e2e_test.rs:141-146: `mae <= 1.0` against ONNX Runtime's logits).
SV_MIN_AGREEMENT, SV_MAX_MAE, SV_MAX_MAE_OVER_RMS, SV_FLOOR_FACTOR = 0.85, 1.0, 0.06, 1.5


@pytest.mark.gpu
def test_c2_c3_sensevoice_shaped_graph_against_the_oracle_forward(ctx):
    """VERDICT r5 "missing 2": the WHOLE 70-layer recogniser, device against oracle, end to end -- configs[2] (one 30 s utterance, the
    plan the bench times, compiled with every fused form incl. round 6's half-layer statements) against oracle/plan_ref.py running the
    SAME plan statement by statement, and utterances 0 and 31 of a configs[3] shard (batch 32 on the device) against the oracle's
    forward of those two utterances.  Judged as the reference judges this model (examples/sensevoice/tests/e2e_test.rs:126-189: MAE
    of the logits -- its bar is 1.0 -- and arg-max agreement per frame), for the shipped attention AND for LELE_HIP_ATTENTION_EXACT=1.

    What such a comparison CAN show.  Every linear re-quantises its input to 8 bits with a range taken from the data: two inputs a
    relative d apart get different codes on a fraction ~ d / (step / sigma) of the elements, each by a whole step, so the outputs part
    by ~ 0.2 sqrt(d) -- a last-bit difference (1e-7) becomes 1e-4 after one linear, 1e-2 after three, and saturates at the layer's
    quantisation noise.  With the plain section-8(d) draws every branch is larger than the residual stream and the saturated noise
    of 140 half-layers decorrelates two bit-careful implementations completely (20 % equal arg-max ids, round 5).  With damped
    branches (tools/sensevoice_graph.py) the noise a branch adds is a few % of a few % of the stream and the forward is comparable:
    but no damping takes it below ~2 % of the logits' rms (measured down to branches of 1 % of the stream) -- the final LayerNorm
    and the CTC linear re-quantise once more.  So the bars are: the reference's MAE bar, arg-max agreement, and -- the sharp one --
    the device is no farther from the oracle than its two attention paths (1e-6 apart at every attention output, everything else
    bit-identical) are from EACH OTHER: the distance is the graph's own sensitivity, not an error of either side."""
    import json
    from oracle import plan_ref
    from sensevoice_graph import Encoder
    from tests.test_fullsize_graph import features
    enc = Encoder(ctx, 70, damped=True)
    report = {"weights": "SURVEY 8(d) draws, residual branches damped (tools/sensevoice_graph.py DAMP_*)", "bars": {
        "min_argmax_agreement": SV_MIN_AGREEMENT, "max_mae (e2e_test.rs:141)": SV_MAX_MAE, "max_mae_over_rms": SV_MAX_MAE_OVER_RMS,
        "mae_vs_oracle <= this x mae between the device's two attention paths": SV_FLOOR_FACTOR}}
    # ---- configs[2]: the device's plan itself on the oracle
    feats = features(ctx, 1, 30)
    plan, w, dev = _sensevoice_logits(ctx, enc, 1, feats, exact=True)
    ref = plan_ref.run(plan, w, {"feats": feats.numpy()})[0]
    for name, got in dev.items():
        report["configs2 " + name] = logits_agreement(got, ref)
    # ---- configs[3]: batch 32 on the device, utterances 0 and 31 on the oracle (a batch-2 plan of the same encoder: per-utterance
    #      dynamic quantisation makes an utterance's forward independent of its batch)
    feats32 = features(ctx, 32, 10)
    _plan32, _w32, dev32 = _sensevoice_logits(ctx, enc, 32, feats32, exact=True)
    from lele_amd.compiler import compile_model
    from lele_amd.plan import load_weights_bin
    from sensevoice_graph import encoder_onnx
    plan2, blob2 = compile_model(encoder_onnx(enc, 2), "sensevoice_shaped")
    ref2 = plan_ref.run(plan2, load_weights_bin(plan2, blob2), {"feats": feats32.numpy()[[0, 31]]})[0]
    for name, got in dev32.items():
        report["configs3 utterances 0, 31 " + name] = logits_agreement(got[[0, 31]], ref2)
    report["configs3 shipped vs exact (all 32)"] = logits_agreement(dev32["shipped"], dev32["exact"])
    report["configs2 shipped vs exact"] = logits_agreement(dev["shipped"], dev["exact"])
    print("\nSENSEVOICE_GRAPH_ORACLE " + json.dumps(report))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "sensevoice_graph_oracle.json"), "w"), indent=1)
    floor = {"configs2": report["configs2 shipped vs exact"]["mae"], "configs3": report["configs3 shipped vs exact (all 32)"]["mae"]}
    for key, r in report.items():
        if not key.startswith("configs") or "vs exact" in key:
            continue
        assert r["argmax_agreement"] >= SV_MIN_AGREEMENT, (key, r)
        assert r["mae"] <= SV_MAX_MAE and r["mae_over_rms"] <= SV_MAX_MAE_OVER_RMS, (key, r)
        assert r["mae"] <= SV_FLOOR_FACTOR * floor[key.split()[0]], (key, r, floor)
