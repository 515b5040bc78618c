import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests.test_compiler import Seq, export
from lele_amd.compiler import compile_model
from lele_amd.plan import Runner, load_weights_bin
from lele_amd.tensor import TensorView
import lele_amd
ctx = lele_amd._lib.Ctx(0)
s = Seq()
plan, blob = compile_model(export(s, (torch.randn(1, 8, 12),), opset=17, dynamic_axes={"x": {2: "t"}}))
x = torch.randn(1, 8, 12, generator=torch.Generator().manual_seed(12))
y0 = torch.tanh(s.conv(x)).permute(2, 0, 1)
y1, _ = s.lstm(y0)
y2, _ = s.gru(y1)
want = {"_Transpose_output_0": y0, "_lstm_Squeeze_output_0": y1, "_gru_Squeeze_output_0": y2, "y": s(x)}
for name, w in want.items():
    idx = max(i for i, st in enumerate(plan["statements"]) if name in st["out"])
    p2 = dict(plan, outputs=[name], statements=plan["statements"][:idx + 1])
    r = Runner(p2, load_weights_bin(p2, blob), ctx)
    got = r.run({"x": TensorView(ctx.buf().upload(x.numpy()))})[0].numpy()
    w = w.detach().numpy()
    print(name, got.shape, w.shape, float(np.abs(got.reshape(w.shape) - w).max()))
