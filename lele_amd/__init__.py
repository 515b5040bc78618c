"""lele_amd -- MI355X (gfx950) implementation of miuda-ai/lele's hot path behind lele's own interface.

Layout (host mirror of the reference crate):
    lele_amd.tensor.TensorView            <-> lele::tensor::TensorView           (src/tensor.rs)
    lele_amd.features.{SenseVoiceFrontend, FeatureConfig, Cmvn, Lfr, RealFft}
                                          <-> lele::features::*                  (src/features)
    lele_amd.kernels.*                    <-> lele::kernels::*                   (src/kernels)
Everything computes in liblele_hip.so (hand-written HIP, C ABI in include/lele_hip.h).  There is no CPU path.
"""
from . import _lib
from ._lib import LeleError

_default_ctx = None


def default_ctx(device=0):
    """One LeleCtx per process by default (lele is single-threaded; one ctx per host thread)."""
    global _default_ctx
    if _default_ctx is None or _default_ctx.device != device:
        _default_ctx = _lib.Ctx(device)
    return _default_ctx


def set_default_ctx(ctx):
    global _default_ctx
    _default_ctx = ctx


from .tensor import TensorView  # noqa: E402
from . import features, kernels  # noqa: E402,F401

__all__ = ["TensorView", "features", "kernels", "default_ctx", "set_default_ctx", "LeleError"]
