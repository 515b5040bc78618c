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
