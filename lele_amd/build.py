"""Build liblele_hip.so (gfx950) in-tree with hipcc.  No torch, no cmake: `python -m lele_amd.build`.

-ffp-contract=off on every translation unit: parity with the reference is defined roundings-first, so an
a*b+c is fused only where the source says fma (mirroring where lele's AVX2 code says _mm256_fmadd_ps).
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# LELE_HIP_LAB=1 builds the developer's library instead: the same sources with -DLELE_HIP_LAB (in-kernel cycle stamps, ablation
# switches, experimental kernels), as liblele_hip_lab.so -- never loaded unless LELE_HIP_LAB=1 is set at import (lele_amd/_lib.py)
LAB = os.environ.get("LELE_HIP_LAB", "0") not in ("", "0")
# LELE_HIP_DEBUG_BOUNDS=1 builds liblele_hip_dbg.so: the same sources with -DLELE_HIP_DEBUG_BOUNDS, in which the loaders and the
# epilogues ASSERT their coordinates on the device (common.h, LELE_DEV_ASSERT: a violation prints where it happened and traps the
# kernel).  Loaded with LELE_HIP_LIBRARY=liblele_hip_dbg.so; tests/test_debug_bounds.py runs parity cases on it.
DBG = os.environ.get("LELE_HIP_DEBUG_BOUNDS", "0") not in ("", "0")
OBJ = os.path.join(HERE, "_build_dbg" if DBG else "_build_lab" if LAB else "_build")
LIB = os.path.join(HERE, "liblele_hip_dbg.so" if DBG else "liblele_hip_lab.so" if LAB else "liblele_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=" + ARCH, "-Wall",
         "-Wno-unused-function", "-Wno-unused-variable", "-Wno-unused-result"] + (["-DLELE_HIP_LAB=1"] if LAB else []) + \
        (["-DLELE_HIP_DEBUG_BOUNDS=1"] if DBG else [])


# Per-file additions.  frontend.hip: the SLP vectoriser pairs the scalar multiplies of the pre-emphasis into v_pk_mul_f32, which cannot
# take the lane rotation as a DPP operand (25 v_mov_dpp + 25 v_mov a pass come back) -- and a packed f32 instruction issues at
# half rate on gfx950 anyway; the FFT's packed arithmetic is written as float2 and stays packed.
# conv.hip: the same for the convolutions' epilogues and the loaders' piece splits (packed f32 beside the matrix core; round 6: stride-2 window
# kernels -3 ... -5 %, the 3-channel stem -4.7 %, 51 geometries -1.0 %; bit-identical).
FILE_FLAGS = {"frontend.hip": ["-fno-slp-vectorize"], "quant.hip": ["-fno-slp-vectorize"], "conv.hip": ["-fno-slp-vectorize"]}


def hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: liblele_hip.so cannot be built")


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdrs = tuple(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + (
        os.path.join(HERE, "..", "include", "lele_hip.h"),)
    cc = hipcc()
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, hdrs):
            jobs.append([cc] + FLAGS + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout)
        return r.stdout

    with cf.ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if jobs or force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        run([cc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs + ["-ldl"])
    if not LAB and not DBG:
        build_runner(force, run)
    return LIB


RUNNER = os.path.join(HERE, "lele_run")


def build_runner(force=False, run=None):
    """lele_run: the native plan runner (host/lele_run.cpp + plan_runner.hpp + lele.hpp over the C ABI), plain g++"""
    host = os.path.join(HERE, "host")
    deps = [os.path.join(host, f) for f in ("lele_run.cpp", "plan_runner.hpp", "lele.hpp")] + [os.path.join(HERE, "..", "include", "lele_hip.h")]
    if not force and os.path.exists(RUNNER) and all(os.path.getmtime(d) <= os.path.getmtime(RUNNER) for d in deps) \
            and os.path.getmtime(LIB) <= os.path.getmtime(RUNNER):
        return RUNNER
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(HERE, "..", "include"), "-I", host, deps[0], "-L", HERE, "-llele_hip",
           "-Wl,-rpath,$ORIGIN", "-o", RUNNER]
    if run is None:
        subprocess.check_call(cmd)
    else:
        run(cmd)
    return RUNNER


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
