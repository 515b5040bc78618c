"""The bounds-asserting build of the library (LELE_HIP_DEBUG_BOUNDS=1 python -m lele_amd.build -> liblele_hip_dbg.so, SURVEY.md
section 5: "bounds-asserting debug kernels"): the GEMM core's loaders, the window kernels' LDS offsets and the epilogues' store
coordinates assert what they assume (lele_amd/csrc/common.h, LELE_DEV_ASSERT).  The parity cases of the matrix products and the
convolutions -- ragged shapes, views, every route -- run on it in a child process: a violated assertion would trap the kernel and fail
the run.  CPU part: the library is a different build that really carries the assertions."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DBG = os.path.join(ROOT, "lele_amd", "liblele_hip_dbg.so")


def _dbg_lib():
    # the build is incremental (nothing happens when the library is newer than every source): never test a stale one
    env = dict(os.environ, LELE_HIP_DEBUG_BOUNDS="1")
    subprocess.check_call([sys.executable, "-m", "lele_amd.build"], cwd=ROOT, env=env, stdout=subprocess.DEVNULL)
    return DBG


def test_debug_library_carries_the_assertions_and_the_whole_abi():
    import ctypes as C
    from lele_amd import _lib
    data = open(_dbg_lib(), "rb").read()
    assert b"lele_hip bounds assertion failed" in data
    assert b"lele_hip bounds assertion failed" not in open(_lib.LIB_PATH, "rb").read()   # the product library has none of it
    lib = C.CDLL(DBG)
    assert not [n for n in _lib.exported_symbols() if not hasattr(lib, n)]


@pytest.mark.gpu
def test_parity_cases_run_clean_on_the_bounds_asserting_build():
    _dbg_lib()
    env = dict(os.environ, LELE_HIP_LIBRARY="liblele_hip_dbg.so")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gemm.py", "tests/test_conv_rnn.py", "tests/test_channel_views.py", "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = r.stdout[-3000:]
    assert r.returncode == 0 and "bounds assertion failed" not in r.stdout, tail
    assert " passed" in tail
