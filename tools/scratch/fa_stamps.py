import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
H, DH = 4, 128
QC = [["slice", 2, 0, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]
KC = [["slice", 2, 512, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 3, 1]]]
VC = [["slice", 2, 1024, 512], ["reshape", [0, 0, H, DH]], ["transpose", [0, 2, 1, 3]]]
import torch
from lele_amd import kernels as K
from lele_amd._lib import Ctx, Weight
ctx = Ctx()
rng = np.random.default_rng(0)
scale = Weight(np.array([DH ** -0.5], np.float32))
b, t = 32, 171
qd = ctx.buf().upload((rng.standard_normal((b, t, 1536)) * 1.5).astype(np.float32))
dst = ctx.buf()
call = lambda: K.attention_view(qd, QC, qd, KC, qd, VC, scale, [0, 2, 1, 3], [0, 0, H * DH], out=dst, ctx=ctx)
for _ in range(3):
    call()
ctx.sync()
nwg = b * H * 2
dbg = torch.zeros((nwg, 8, 64), dtype=torch.int64, device="cuda")
torch.cuda.synchronize()
os.environ["LELE_HIP_ATTN_STAMPS"] = hex(dbg.data_ptr())
call()
ctx.sync()
tt = dbg.cpu().numpy().astype(np.float64)
t0 = tt[:, :, 0].min(axis=1, keepdims=True)
np.set_printoptions(linewidth=220, suppress=True)
for wg in (0, 8, 100):
    print("workgroup", wg)
    for w in range(8):
        r = tt[wg, w] - t0[wg]
        r[tt[wg, w] == 0] = -1
        n = 31 if w < 4 else 24
        print("  wave", w, " ".join("%6d" % v for v in r[:n]), "end %d" % r[63])
# medians over workgroups with 4 live waves (qb = 0)
full = [wg for wg in range(nwg) if tt[wg, 3, 3] > 0]
c = tt[full][:, 0] - t0[full]
pr = tt[full][:, 4] - t0[full]
print("median consumer wave 0:", " ".join("%6d" % v for v in np.median(c, axis=0)[:31]), "end %d" % np.median(c[:, 63]))
print("median producer wave 4:", " ".join("%6d" % v for v in np.median(pr, axis=0)[:24]))
pr = tt[full][:, 6] - t0[full]
print("median producer wave 6:", " ".join("%6d" % v for v in np.median(pr, axis=0)[:24]))
