python tools/silero_graph.py --runs 2 2>&1 | tail -2
python - <<'PY'
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import numpy as np, silero_graph as S
import lele_amd
from lele_amd import kernels as K
from lele_amd.compiler import compile_model
from lele_amd.plan import Runner, load_weights_bin
from lele_amd.tensor import TensorView
plan, blob = compile_model(S.build_onnx(), "s")
ctx = lele_amd._lib.Ctx(0)
r = Runner(plan, load_weights_bin(plan, blob), ctx)
rng=np.random.default_rng(0)
xs=[(0.3*rng.standard_normal((1,576))).astype(np.float32) for _ in range(4)]
xb,hb,cb=ctx.buf(),ctx.buf(),ctx.buf()
z=np.zeros((1,1,128),np.float32)
def run_eager():
    h=TensorView(hb.upload(z)); c=TensorView(cb.upload(z)); out=[]
    for x in xs:
        xv=TensorView(xb.upload(x))
        p,hn,cn=r.run({"x":xv,"sr":np.array([16000]),"h0":h,"c0":c})
        out.append((p.numpy().copy(), hn.numpy().copy(), cn.numpy().copy()))
        K.view_copy(hn,[],out=hb,ctx=ctx); K.view_copy(cn,[],out=cb,ctx=ctx)
    return out
a=run_eager(); b=run_eager()
print("eager repeat equal", all(np.array_equal(u,v) for x,y in zip(a,b) for u,v in zip(x,y)))
print([float(q[0].reshape(-1)[0]) for q in a])
h=TensorView(hb.upload(z)); c=TensorView(cb.upload(z)); xv=TensorView(xb.upload(xs[0]))
ctx.sync(); ctx.graph_begin()
p,hn,cn=r.run({"x":xv,"sr":np.array([16000]),"h0":h,"c0":c})
K.view_copy(hn,[],out=hb,ctx=ctx); K.view_copy(cn,[],out=cb,ctx=ctx)
g=ctx.graph_end()
hb.upload(z); cb.upload(z)
for i,x in enumerate(xs):
    xb.upload(x); g.launch(); ctx.sync()
    print(i, float(p.numpy().reshape(-1)[0]), np.array_equal(hn.numpy(), a[i][1]), np.array_equal(cn.numpy(), a[i][2]), np.abs(hn.numpy()-a[i][1]).max())
PY
