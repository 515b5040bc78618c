"""the CPU restatement under AddressSanitizer / UndefinedBehaviorSanitizer.

`make -C oracle liboracle_asan.so` builds every oracle source with -fsanitize=address,undefined (-fno-sanitize-recover: the first
finding aborts); a child interpreter with libasan preloaded and ORACLE_SANITIZE=1 (oracle/pyoracle.py then loads that library)
replays the golden-vector tests, the fast-loop tests and a set of edge shapes -- ragged tails, one-element tensors, windows that
hang over every edge -- through it.  The oracle is the checker of every parity claim; an out-of-bounds read in it would pin the
device kernels to garbage."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

EDGE = r'''
import numpy as np
from oracle import pyoracle as O, npref
assert O.SANITIZE and O.build().endswith("liboracle_asan.so")
rng = np.random.default_rng(0)
for n in (400, 401, 559, 560, 1599, 16000):                       # front-end: the shortest utterances and ragged frame counts
    O.frontend_compute(rng.standard_normal(n).astype(np.float32) * 0.1)
for n in (8, 64, 512, 1024):
    O.rfft(rng.standard_normal(n).astype(np.float32))
for m, k, n in ((1, 1, 1), (3, 7, 5), (17, 33, 9), (2, 31, 64)):    # quantised linear: K off the 8- and 32-lane grids
    x = rng.standard_normal((2, m, k)).astype(np.float32)
    w = np.clip(np.round(128 + 60 * rng.standard_normal((k, n))), 0, 255).astype(np.float32)
    O.fused_quantized_linear(x, w, np.full(n, 0.01, np.float32), np.array([128.0], np.float32), np.zeros(n, np.float32), True)
    O.matmul(x, rng.standard_normal((k, n)).astype(np.float32), acc32=True)
for ln in (1, 7, 8, 9, 31, 513):                                   # 8-wide bodies with scalar tails
    v = rng.standard_normal((3, ln)).astype(np.float32)
    O.softmax(v)
    O.layer_norm(v, np.ones(ln, np.float32), np.zeros(ln, np.float32))
    for name in ("exp", "sigmoid", "silu", "erf", "gelu", "tanh"):
        O.unary(name, v)
for (c, h, w_, oc, k, s, g, pads) in ((1, 1, 1, 1, 1, 1, 1, [0] * 4), (3, 5, 7, 4, 3, 2, 1, [1] * 4), (4, 6, 6, 4, 3, 1, 4, [1] * 4),
                                      (2, 4, 9, 3, 5, 1, 1, [2, 2, 2, 2]), (6, 7, 5, 6, 3, 1, 2, [0, 1, 2, 0])):
    x = rng.standard_normal((2, c, h, w_)).astype(np.float32)
    w = rng.standard_normal((oc, c // g, k, k)).astype(np.float32)
    for act in (None, "relu", "silu"):
        O.conv2d(x, w, None, [1, 1], g, pads, [s, s], act)
        O.conv2d_im2col(x, w, np.zeros(oc, np.float32), [1, 1], g, pads, [s, s], act)
O.conv_transpose(rng.standard_normal((1, 3, 4, 5)).astype(np.float32), rng.standard_normal((3, 2, 2, 2)).astype(np.float32), None, [1, 1], 1, [0] * 4, [2, 2])
O.lstm(rng.standard_normal((3, 1, 5)).astype(np.float32), rng.standard_normal((1, 16, 5)).astype(np.float32), rng.standard_normal((1, 16, 4)).astype(np.float32))
O.gru(rng.standard_normal((3, 1, 5)).astype(np.float32), rng.standard_normal((1, 12, 5)).astype(np.float32), rng.standard_normal((1, 12, 4)).astype(np.float32))
O.cmvn(rng.standard_normal((1, 560)).astype(np.float32))
print("edge shapes clean")
'''


def test_oracle_under_asan_and_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE, text=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), "libasan.so not found next to gcc"
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "liboracle_asan.so"])
    env = dict(os.environ, ORACLE_SANITIZE="1", LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", EDGE], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "edge shapes clean" in r.stdout, r.stdout[-4000:]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_oracle_golden.py", "tests/test_oracle_fast.py"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
