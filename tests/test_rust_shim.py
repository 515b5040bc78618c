"""The generated Rust shim crate (rust/lele-hip, a crate named `lele`) -- what can be checked without a Rust toolchain:

  * the committed files are exactly what tools/rust_shim/gen.py emits from include/lele_hip.h + signatures.json (not hand-edited, not stale);
  * every `lele::kernels::<fn>` that lele's emitter writes, that appears in lele's generated Yolo26n-seg source, and that this
    repository's own compiler emits as a lele kernel has a `pub fn` of that name in src/kernels.rs, carrying lele's signature text;
  * every C symbol the crate calls is declared in src/ffi.rs, and src/ffi.rs declares exactly the header's symbols."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CRATE = os.path.join(ROOT, "rust", "lele-hip", "src")


def test_generated_files_are_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rust_shim", "gen.py"), "--check"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr + r.stdout


def kernel_fns():
    txt = open(os.path.join(CRATE, "kernels.rs")).read()
    fns = set(re.findall(r"(?m)^pub fn (\w+)", txt))
    fns |= set(re.findall(r"(?m)^pub use self::\w+ as (\w+);", txt))
    return fns, txt


def test_every_emitted_kernel_name_exists_with_leles_signature():
    fns, txt = kernel_fns()
    names = json.load(open(os.path.join(ROOT, "tests", "golden", "generated_kernel_names.json")))
    not_kernels = {"utils", "argmax"}   # a module path; a name upstream's emitter writes but src/kernels never defines
    missing = [n for n in names["emitter"] + names["yolo26seg"] if n not in fns and n not in not_kernels]
    assert not missing, missing
    # this repository's compiler: every call it can emit is either a lele kernel (must be in the shim) or one of its own fused forms
    lower = open(os.path.join(ROOT, "lele_amd", "compiler", "lower.py")).read()
    emitted = set(re.findall(r'self\.emit\([^,]+,\s*"([a-z_0-9]+)"', lower))
    own = {"add3", "depthwise_conv1d_tlc", "halves_pow_add_sqrt", "matmul_view", "view_copy", "softmax_scaled", "attention_view",
           "fused_quantized_linear_residual", "cast_to_i64"}
    assert not [n for n in emitted - own if n not in fns]
    # signature text: exactly lele's declaration
    sigs = json.load(open(os.path.join(ROOT, "tools", "rust_shim", "signatures.json")))["functions"]
    for f in sigs:
        if not f["exported"]:
            continue
        decl = "pub fn %s%s(%s)%s {" % (f["name"], f["generics"], ", ".join(f["params"]), (" -> " + f["ret"]) if f["ret"] else "")
        assert decl in txt, f["name"]


def test_ffi_block_matches_the_header():
    from lele_amd._lib import exported_symbols
    ffi = open(os.path.join(CRATE, "ffi.rs")).read()
    declared = set(re.findall(r"pub fn (lele_hip_\w+)\(", ffi))
    assert declared == set(exported_symbols())
    used = set()
    for name in ("kernels.rs", "rt.rs", "features.rs", "tensor.rs"):
        used |= set(re.findall(r"ffi::(lele_hip_\w+)", open(os.path.join(CRATE, name)).read()))
    assert used <= declared, sorted(used - declared)
    assert len(used) >= 50   # the operator library, not a sample of it
