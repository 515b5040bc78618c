"""The per-lane FFT code the HIP kernel runs (lele_amd/csrc/fe_core.h) is compiled for the host and driven by a
16-lane emulator (tests/emu/fe_emulate.cpp): its 257 bins must equal the oracle's restatement of the reference
FFT BIT FOR BIT, for both the generic and the zero-folded phase A."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    src = os.path.join(HERE, "emu", "fe_emulate.cpp")
    so = os.path.join(HERE, "emu", "libfe_emu.so")
    hdr = os.path.join(HERE, "..", "lele_amd", "csrc", "fe_core.h")
    if not os.path.exists(so) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-mfma", "-fPIC", "-shared", "-o", so,
                               src])
    return C.CDLL(so)


def test_lane_mapping_is_bit_exact(emu, orc):
    twr, twi, _ = orc.precompute_twiddles(512)
    twr, twi = np.ascontiguousarray(twr), np.ascontiguousarray(twi)
    assert twr[0] == 1.0 and twi[0] == 0.0 and twr[1] == 1.0 and twi[1] == 0.0  # what phase_a_fast folds away
    rng = np.random.default_rng(0)
    for trial in range(64):
        fr = np.zeros(512, np.float32)
        fr[:400] = (rng.standard_normal(400) * 10 ** rng.uniform(-3, 4)).astype(np.float32)
        if trial % 3 == 0:
            fr[:400] += np.float32(1e4) * np.sin(0.3 * np.arange(400)).astype(np.float32)
        re, im = orc.rfft(fr, 2)
        for fast in (0, 1):
            ore, oim = np.empty(257, np.float32), np.empty(257, np.float32)
            emu.fe_emulate_fft512(fr.ctypes.data_as(C.c_void_p), twr.ctypes.data_as(C.c_void_p),
                                  twi.ctypes.data_as(C.c_void_p), ore.ctypes.data_as(C.c_void_p),
                                  oim.ctypes.data_as(C.c_void_p), C.c_int(fast))
            assert np.array_equal(ore, re) and np.array_equal(oim, im), (trial, fast)
