"""sha256 of the front-end's outputs on seeded PCM (A/B of two library builds inside one gpurun call: LELE_HIP_LIBRARY=...)."""
import hashlib
import numpy as np
from lele_amd import features, tensor
rng = np.random.default_rng(7)
n = np.arange(480000)
h = hashlib.sha256()
fe = features.SenseVoiceFrontend()
for amp in (1.0, 1e-3, 3e-5):
    pcm = (amp * (0.3 * np.sin(2 * np.pi * 220 * n / 16000) + 0.2 * np.sin(2 * np.pi * 1000 * n / 16000)
                  + 0.05 * rng.uniform(-1, 1, (4, n.size)))).astype(np.float32)
    out = fe.compute_batch(pcm)
    h.update(np.ascontiguousarray(out.numpy() if hasattr(out, "numpy") else np.asarray(out)).tobytes())
    lm = fe.logmel(pcm[0, :160000])
    h.update(np.ascontiguousarray(lm.numpy() if hasattr(lm, "numpy") else np.asarray(lm)).tobytes())
print(h.hexdigest())
