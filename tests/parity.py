"""The f32 parity bar of BASELINE.json ("mel and NN ops within 1e-4 relative f32"), in one place.

close_f32: element by element, |got - want| <= tol * |want| + tol * rms(want) + 1e-7.  The rms term is the round-off floor of a sum
of products: an output that cancels to ~0 carries the error of its terms, which scale with the tensor's typical magnitude.  No
max(1, .) slack and no scaling by the tensor's largest element."""
import numpy as np


def close_f32(got, want, tol=1e-4, what=""):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    if not want.size:
        return
    rms = float(np.sqrt(np.mean(np.square(want))))
    bad = np.abs(got - want) > tol * np.abs(want) + tol * rms + 1e-7
    assert not bad.any(), "%s: %d of %d elements outside %g (max abs diff %.3e, rms %.3e)" % (
        what, int(bad.sum()), want.size, tol, float(np.abs(got - want).max()), rms)
