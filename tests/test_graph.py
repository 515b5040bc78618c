"""hipGraph capture of an op sequence (include/lele_hip.h, lele_hip_graph_*): a replay must produce exactly what the
eager calls produce on the current contents of the same device buffers, and ops that cannot be recorded fail loudly."""
import numpy as np
import pytest


@pytest.mark.gpu
def test_graph_replay_matches_eager_and_tracks_buffer_contents(ctx):
    import lele_amd
    from lele_amd import kernels as K
    from lele_amd._lib import Weight
    rng = np.random.default_rng(0)
    a_buf = ctx.buf()
    a = a_buf.upload(rng.standard_normal((4, 40, 32)).astype(np.float32))
    w = Weight(rng.standard_normal((32, 48)).astype(np.float32))
    bias = Weight(rng.standard_normal(48).astype(np.float32))
    o1, o2, o3 = ctx.buf(), ctx.buf(), ctx.buf()

    def seq():
        y = K.matmul_fused_add(a, w, bias, out=o1, ctx=ctx)
        y = K.silu(y, out=o2, ctx=ctx)
        return K.softmax(y, -1, out=o3, ctx=ctx)

    eager = seq().numpy()  # also warms the weight cache and sizes the buffers
    ctx.graph_begin()
    res = seq()
    g = ctx.graph_end()
    g.launch()
    shape = res.shape
    assert np.array_equal(o3.to_numpy(shape), eager)  # read the buffer itself: TensorView caches its host copy
    # new input in the same buffer: the replay follows it, eager agrees bit for bit
    a_buf.upload(rng.standard_normal((4, 40, 32)).astype(np.float32))
    g.launch()
    replay = o3.to_numpy(shape)
    assert not np.array_equal(replay, eager)
    assert np.array_equal(seq().numpy(), replay)
    g.close()


@pytest.mark.gpu
def test_graph_capture_rejects_host_inputs_and_growth(ctx):
    import lele_amd
    from lele_amd import kernels as K
    x = np.ones((8, 8), np.float32)
    ctx.graph_begin()
    with pytest.raises(lele_amd.LeleError, match="graph capture"):
        K.matmul(x, x, ctx=ctx)  # pageable host inputs cannot be recorded
    ctx.graph_abort()
    d = ctx.buf().upload(x)
    small = ctx.buf()
    ctx.graph_begin()
    with pytest.raises(lele_amd.LeleError, match="graph capture"):
        K.matmul(d, d, out=small, ctx=ctx)  # the output buffer was never sized: growing it would allocate
    ctx.graph_abort()
    # the ctx is usable again after an aborted capture
    assert np.array_equal(K.matmul(d, d, out=small, ctx=ctx).numpy(), x @ x)
