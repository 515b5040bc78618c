"""depthwise_conv1d_tlc at the configs[3] shard shape (V of [32, 171, 1536], k = 11, residual): graph-replayed time."""
import sys, time, numpy as np
sys.path.insert(0, ".")
import lele_amd
from lele_amd import kernels as K
from lele_amd.tensor import TensorView
ctx = lele_amd._lib.Ctx(0)
rng = np.random.default_rng(0)
x = rng.standard_normal((32, 171, 1536)).astype(np.float32)
w = rng.standard_normal((512, 1, 11)).astype(np.float32)
xd = TensorView(ctx.buf().upload(x)); wd = lele_amd._lib.Weight(w)
out = ctx.buf()
def run(): return K.depthwise_conv1d_tlc(xd, wd, None, 5, 5, x_offset=1024, add_input=True, out=out, ctx=ctx)
y = run(); ctx.sync()
ctx.graph_begin()
for _ in range(50): run()
g = ctx.graph_end()
for _ in range(3): g.launch()
ctx.sync(); t0 = time.perf_counter()
for _ in range(20): g.launch()
ctx.sync(); dt = (time.perf_counter() - t0) / (20 * 50)
import hashlib
print("%.2f us per call, sha %s" % (dt * 1e6, hashlib.sha256(y.numpy().tobytes()).hexdigest()[:16]))
