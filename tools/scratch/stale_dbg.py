import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import numpy as np
import lele_amd
from lele_amd import kernels as K
from lele_amd._lib import Weight
from lele_amd.tensor import TensorView
ctx = lele_amd._lib.Ctx(0)
rng=np.random.default_rng(0)
def check(name, fn, shapes):
    bufs=[ctx.buf() for _ in shapes]
    vals=[(0.3*rng.standard_normal(s)).astype(np.float32) for s in shapes]
    tv=[TensorView(b.upload(v)) for b,v in zip(bufs,vals)]
    outs=fn(*tv)                       # eager once
    outs=outs if isinstance(outs,(list,tuple)) else [outs]
    ctx.sync(); ctx.graph_begin()
    o2=fn(*tv); o2=o2 if isinstance(o2,(list,tuple)) else [o2]
    g=ctx.graph_end()
    g.launch(); ctx.sync()
    first=[o.numpy().copy() for o in o2]
    for b,s in zip(bufs,shapes): b.upload((0.3*rng.standard_normal(s)).astype(np.float32))
    want=[o.numpy().copy() for o in (lambda r: r if isinstance(r,(list,tuple)) else [r])(fn(*tv))]
    for b,s,v in zip(bufs,shapes,vals): pass
    g.launch(); ctx.sync()
    second=[o.numpy().copy() for o in o2]
    print(name, "stale" if all(np.array_equal(a,b) for a,b in zip(first,second)) else "fresh", "match-eager" if all(np.array_equal(a,b) for a,b in zip(second,want)) else "MISMATCH")
w=Weight((0.1*rng.standard_normal((258,1,256))).astype(np.float32))
ob=[ctx.buf() for _ in range(8)]
two=Weight(np.array([2.0],np.float32))
check("conv1d_stft", lambda x: K.conv1d(K.unsqueeze(x,[1]), w, None, [1], 1, [0,0], [128], out=ob[0], ctx=ctx), [(1,576)])
W=Weight((0.1*rng.standard_normal((1,512,128))).astype(np.float32)); R=Weight((0.1*rng.standard_normal((1,512,128))).astype(np.float32)); B=Weight((0.1*rng.standard_normal((1,1024))).astype(np.float32))
import inspect
print(inspect.signature(K.lstm))
check("lstm", lambda x,h,c: K.lstm(x, W, R, B, None, h, c, outs=[ob[1],ob[2],ob[3]], ctx=ctx), [(1,1,128),(1,1,128),(1,1,128)])
check("slice_pow", lambda x: getattr(K,"pow")(K.slice(x,[0],[129],[1],[1],out=ob[4],ctx=ctx), two, out=ob[5], ctx=ctx), [(1,258,3)])
check("reduce_mean", lambda x: K.reduce_mean(x,[2],False,out=ob[6],ctx=ctx), [(1,1,3)])
check("view_copy", lambda x: K.view_copy(x,[],out=ob[7],ctx=ctx), [(1,1,128)])
