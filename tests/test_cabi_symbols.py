"""The C-ABI library loads on a CPU-only box and exports every symbol include/lele_hip.h declares
(no compute calls here: there is no GPU)."""
import ctypes as C
import os


def test_library_exports_every_declared_symbol():
    from lele_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from lele_amd import build
        build.build()
    lib = C.CDLL(_lib.LIB_PATH)
    names = _lib.exported_symbols()
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, "declared in include/lele_hip.h but not exported: %s" % missing


def test_last_error_is_callable_without_gpu():
    from lele_amd import _lib
    assert isinstance(_lib.lib().lele_hip_last_error(), bytes)


def test_table_builders_are_the_oracles_tables():
    """lele_hip_hann_window / lele_hip_mel_filterbank / the HTK mel scale: host arithmetic behind the C ABI (no GPU needed), what the
    Rust binding's `lele::features::{hann_window, mel_filterbank, ..}` return -- bit for bit the oracle's restatement of
    window.rs:2-13 and mel.rs:1-56, pinned against lele's own test values in tests/test_oracle_golden.py"""
    import numpy as np
    from lele_amd import _lib
    from oracle import pyoracle as O
    lib = _lib.lib()
    lib.lele_hip_hz_to_mel_htk.restype = C.c_float
    lib.lele_hip_mel_to_hz_htk.restype = C.c_float
    for n in (0, 1, 2, 7, 400, 512):
        w = np.full(n, 7.0, np.float32)
        assert lib.lele_hip_hann_window(C.c_int64(n), w.ctypes.data_as(C.c_void_p)) == 0
        assert np.array_equal(w, np.asarray(O.hann_window(n), np.float32)), n
    for hz in (0.0, 20.0, 700.0, 1000.0, 8000.0):
        assert np.float32(lib.lele_hip_hz_to_mel_htk(C.c_float(hz))) == np.float32(O.hz_to_mel_htk(hz))
        mel = float(np.float32(O.hz_to_mel_htk(hz)))
        assert np.float32(lib.lele_hip_mel_to_hz_htk(C.c_float(mel))) == np.float32(O.mel_to_hz_htk(mel))
    for sr, n_fft, n_mels, f_min, f_max in ((16000.0, 512, 80, 20.0, None), (16000.0, 400, 23, 0.0, 7600.0), (8000.0, 256, 40, 50.0, None)):
        bank = np.full((n_mels, n_fft // 2 + 1), -1.0, np.float32)
        rc = lib.lele_hip_mel_filterbank(C.c_float(sr), C.c_int64(n_fft), C.c_int64(n_mels), C.c_float(f_min), C.c_int32(f_max is not None),
                                         C.c_float(f_max or 0.0), bank.ctypes.data_as(C.c_void_p))
        assert rc == 0
        want = np.asarray(O.mel_filterbank(sr, n_fft, n_mels, f_min, f_max), np.float32).reshape(bank.shape)
        assert np.array_equal(bank, want), (sr, n_fft, n_mels)
