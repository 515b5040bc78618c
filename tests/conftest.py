import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (test infrastructure): ctypes binding of oracle/liboracle.so, built on demand."""
    from oracle import pyoracle
    pyoracle.lib()
    return pyoracle


@pytest.fixture(scope="session")
def ctx():
    """A LeleCtx on cuda:0.  GPU tests fail loudly (no skip, no fallback) if the HIP library cannot run."""
    # Some tests import torch (its exporter), some bring up an RCCL communicator through the C library (dlopen("librccl.so")).
    # torch ships its own librccl: whichever copy is loaded first must stay the only one in the process (two copies abort the
    # interpreter at exit), so torch -- when present -- goes first, as it does in bench.py.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    import lele_amd
    return lele_amd.default_ctx(0)


def synth_pcm(n, seed=0):
    """SURVEY.md section 8(d) synthetic PCM: two tones + uniform noise"""
    import numpy as np
    rng = np.random.default_rng(seed)
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 220 * t) + 0.2 * np.sin(2 * np.pi * 1000 * t) + 0.05 * rng.uniform(-1, 1, n)
    return x.astype(np.float32)
