"""SenseVoice-shaped encoder, composed operator by operator from the CPU oracle (TEST INFRASTRUCTURE, see oracle.h).

The node sequence is the one `tools/sensevoice_graph.py::Encoder.forward` issues through the C ABI (the assumed topology of
SURVEY.md 8a: SAN-M layers with d=512, 4 x 128 heads, FFN 2048, FSMN depthwise conv k=11, dynamically quantised linears):
every node here is the oracle's restatement of the lele kernel the generated code would call -- `layer_norm`
(norm.rs:226), `fused_quantized_linear` (quantization.rs:77), `conv1d` (conv1d.rs:837), `matmul` (gemm.rs:112),
`softmax` (norm.rs:8), `add`/`mul` (math.rs:414/611), `transpose`/`split`/`reshape` (manipulation.rs).

Used (a) by the `-m gpu` full-size tests as the checker for ONE layer at T=504 / 32x171 on identical inputs (a whole
70-layer stack cannot be compared end to end: a 1-ulp difference flips a u8 rounding of the dynamic quantiser and the
random-weight stack diverges chaotically -- SURVEY.md 7, "end-to-end drift"), and (b) by bench.py's `cpu_baseline` leg as
lele's CPU execution model for the same layer stack (one thread, one kernel call per node).

`weights` are plain numpy arrays (dict per layer); `tools/sensevoice_graph.py::layer_arrays` extracts them from an Encoder.
"""
import numpy as np

from . import pyoracle as O

D, HEADS, DH, FSMN_K = 512, 4, 128, 11


def qlinear(x, p, relu=False):
    """p = (w [K,N] u8-as-f32, scale [N], zero [1], bias [N])"""
    return O.fused_quantized_linear(x, p[0], p[1], p[2], p[3], relu)


def layer_forward(x, L, acc32=True, taps=None):
    """x [B, T, d_in] -> [B, T, 512].  L: dict with ln1, ln2 = (gamma, beta); qkv, out, ffn1, ffn2 = qlinear tuples;
    fsmn [512,1,11]; d_in.  acc32: k-ordered f32 accumulation in the two attention matmuls (the order the device's f32
    MFMA uses; lele's own order inside faer is unpinned).  taps: optional dict that receives intermediates."""
    b, t, _ = x.shape
    xn = O.layer_norm(x, L["ln1"][0], L["ln1"][1], -1, 1e-5)
    qkv = qlinear(xn, L["qkv"])
    q, k, v = qkv[..., :D], qkv[..., D:2 * D], qkv[..., 2 * D:]
    vt = np.ascontiguousarray(v.transpose(0, 2, 1))
    mem = O.conv1d(vt, L["fsmn"], None, [1], D, [FSMN_K // 2, FSMN_K // 2], [1])
    mem = np.ascontiguousarray(mem.transpose(0, 2, 1)) + v
    qh = np.ascontiguousarray(q.reshape(b, t, HEADS, DH).transpose(0, 2, 1, 3))
    kh = np.ascontiguousarray(k.reshape(b, t, HEADS, DH).transpose(0, 2, 3, 1))
    vh = np.ascontiguousarray(v.reshape(b, t, HEADS, DH).transpose(0, 2, 1, 3))
    sc = O.matmul(qh, kh, acc32=acc32) * np.float32(DH ** -0.5)
    pr = O.softmax(sc, -1)
    av = O.matmul(pr, vh, acc32=acc32)
    av = np.ascontiguousarray(av.transpose(0, 2, 1, 3)).reshape(b, t, D)
    att = qlinear(av, L["out"])
    if L["d_in"] == D:
        x1 = (att + mem) + x
    else:
        x1 = att + mem
    xn2 = O.layer_norm(x1, L["ln2"][0], L["ln2"][1], -1, 1e-5)
    h = qlinear(xn2, L["ffn1"], True)
    h2 = qlinear(h, L["ffn2"])
    y = x1 + h2
    if taps is not None:
        taps.update(xn=xn, qkv=qkv, mem=mem, sc=sc, pr=pr, av=av, att=att, x1=x1, h=h, h2=h2)
    return y


def encoder_forward(feats, W, layers=None):
    """feats [B, T, 560] -> logits [B, T+4, vocab]; W: dict(prompt [1,4,560], layers [...], ln_out (g, b), ctc qlinear tuple)"""
    b = feats.shape[0]
    x = np.concatenate([np.broadcast_to(W["prompt"], (b, 4, 560)), feats], axis=1).astype(np.float32)
    for L in (W["layers"] if layers is None else W["layers"][:layers]):
        x = layer_forward(x, L)
    xn = O.layer_norm(x, W["ln_out"][0], W["ln_out"][1], -1, 1e-5)
    return qlinear(xn, W["ctc"])
