import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
import numpy as np, json
import lele_amd
import sensevoice_graph as S
from lele_amd.compiler import compile_model
from lele_amd.plan import Runner, load_weights_bin
from lele_amd.tensor import TensorView
ctx = lele_amd._lib.Ctx(0)
enc = S.Encoder(ctx, layers=1)
feats = np.random.default_rng(2).standard_normal((2, 41, 560)).astype(np.float32)
data = S.encoder_onnx(enc, 2)
res = {}
for fa in (False, True):
    plan, blob = compile_model(data, fuse_attention=fa)
    for name in ("l0_qkv", "l0_avm", "l0_x1"):
        idx = max(i for i, st in enumerate(plan["statements"]) if name in st["out"])
        p2 = dict(plan, outputs=[name], statements=plan["statements"][:idx + 1])
        r = Runner(p2, load_weights_bin(p2, blob), ctx)
        res[fa, name] = r.run({"feats": TensorView(ctx.buf().upload(feats))})[0].numpy()
for name in ("l0_qkv", "l0_avm", "l0_x1"):
    a, b = res[False, name], res[True, name]
    print(name, a.shape, b.shape, float(np.abs(a - b).max()), float(np.abs(a).max()))
a, b = res[False, "l0_avm"], res[True, "l0_avm"]
d = np.abs(a - b)
print(np.unravel_index(d.argmax(), d.shape), d.max(axis=(1, 2)), d.max(axis=(0, 2))[:12], d.reshape(2, 45, 4, 128).max(axis=(0, 1, 3)))
