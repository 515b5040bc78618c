import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, lele_amd
from lele_amd import kernels as K
ctx = lele_amd._lib.Ctx(0)
rng = np.random.default_rng(0)
def dev(a): return ctx.buf().upload(np.ascontiguousarray(a))
def timeit(fn, iters=30):
    out = ctx.buf()
    for _ in range(3): fn(out)
    ctx.sync(); ctx.timer_start()
    for _ in range(iters): fn(out)
    return ctx.timer_stop() / iters
for b, m, k, n in [(128,171,128,171),(128,171,171,128),(1,2048,2048,2048),(1,4096,4096,4096),(1,5472,512,512)]:
    a, bb = dev(rng.standard_normal((b,m,k)).astype(np.float32)), dev(rng.standard_normal((b,k,n)).astype(np.float32))
    ms = timeit(lambda o: K.matmul(a, bb, out=o, ctx=ctx))
    print(b,m,k,n, round(ms*1e3,1), 'us', round(2.0*b*m*k*n/ms/1e9,1), 'TF', flush=True)
