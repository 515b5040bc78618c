"""The C++ host mirror (lele_amd/host/lele.hpp -- the compiled stand-in for the Rust shim of INTEGRATION.md) builds
with a plain host compiler against include/lele_hip.h, fails loudly without a device (CPU), and reproduces the
oracle through the C ABI (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from conftest import synth_pcm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "host_cpp", "host_demo")


def _build():
    libdir = os.path.join(ROOT, "lele_amd")
    assert os.path.exists(os.path.join(libdir, "liblele_hip.so")), "run __graft_entry__.build() first"
    src = os.path.join(ROOT, "tests", "host_cpp", "host_demo.cpp")
    if not os.path.exists(EXE) or os.path.getmtime(EXE) < max(os.path.getmtime(src), os.path.getmtime(
            os.path.join(libdir, "host", "lele.hpp")), os.path.getmtime(os.path.join(libdir, "liblele_hip.so"))):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I",
                               os.path.join(libdir, "host"), src, "-L", libdir, "-llele_hip",
                               "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    return EXE


def test_host_cpp_builds_and_fails_loudly_without_device():
    exe = _build()
    r = subprocess.run([exe, "probe"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("PROBE"), r.stdout + r.stderr


@pytest.mark.gpu
def test_host_cpp_frontend_cmvn_matches_oracle(tmp_path, orc):
    exe = _build()
    pcm = synth_pcm(16000 * 3, seed=4)
    pin, pout, pfeat = tmp_path / "pcm.f32", tmp_path / "out.f32", tmp_path / "feats.f32"
    pcm.astype(np.float32).tofile(pin)
    r = subprocess.run([exe, "run", str(pin), str(pout), str(pfeat)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("OK"), r.stdout + r.stderr
    want = orc.frontend_compute(pcm)
    feats = np.fromfile(pfeat, np.float32).reshape(want.shape)
    # front-end: bit-exact up to the mel sums, ln within 2 ulp -> 1e-6 relative (the contract bar is 1e-4)
    np.testing.assert_allclose(feats, want, rtol=1e-6, atol=2e-6)
    # CMVN is bit-exact on identical input (sequential per-dim sums as cmvn.rs:14-66)
    got = np.fromfile(pout, np.float32).reshape(want.shape)
    assert np.array_equal(got, orc.cmvn(feats))


@pytest.mark.gpu
def test_host_cpp_call_overhead_is_reported():
    exe = _build()
    r = subprocess.run([exe, "latency"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("LATENCY"), r.stdout + r.stderr
    fields = dict(kv.split("=") for kv in r.stdout.split()[1:])
    # the C ABI adds single-digit microseconds per call on the host; the GPU-side floor of a tiny kernel is a few more
    assert float(fields["issue_us_per_call"]) < 50.0 and float(fields["end_to_end_us_per_call"]) < 100.0
    print(r.stdout.strip())
