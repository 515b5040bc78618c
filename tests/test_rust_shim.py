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
    # signature text: exactly lele's declaration (name, generics, parameters, return type), followed by the body or a where clause
    sigs = json.load(open(os.path.join(ROOT, "tools", "rust_shim", "signatures.json")))["functions"]
    for f in sigs:
        if not f["exported"]:
            continue
        head = "pub fn %s%s(%s)%s" % (f["name"], f["generics"], ", ".join(f["params"]), (" -> " + f["ret"]) if f["ret"] else "")
        assert head + " {" in txt or head + "\nwhere\n" in txt, f["name"]
    for name in ("reset_conv_stats", "print_conv_stats"):   # examples/yolo26n-seg/src/main.rs:64,74 (src/kernels/mod.rs:26-29)
        assert "pub fn %s() {" % name in txt


def _type_params(generics):
    inner, out, depth, cur = generics.strip()[1:-1], [], 0, ""
    for ch in inner + ",":
        depth += {"<": 1, "(": 1, ">": -1, ")": -1}.get(ch, 0)
        if ch == "," and depth == 0:
            tok = cur.strip()
            if tok and not tok.startswith("'"):
                out.append(tok.split(":")[0].strip())
            cur = ""
        else:
            cur += ch
    return out


def test_every_generic_forwarder_carries_the_element_bound():
    """Round 2's crate copied lele's bounds verbatim (`T: Clone + Copy + Debug`) onto bodies that call `as_c()` / `rt::*<T: ElementOps>`:
    E0277 on every generic kernel.  Lint, without rustc: every type parameter of every `pub fn` in kernels.rs is bounded by
    ElementOps -- in its generics or in its where clause -- and upstream's own where clauses (gather's I: AsI64, where_op's C) are kept."""
    _, txt = kernel_fns()
    checked = 0
    for m in re.finditer(r"(?m)^pub fn (\w+)\s*(<[^(]*>)?\(([^{]*?)\)([^{]*)\{", txt):
        name, generics, tail = m.group(1), m.group(2) or "", m.group(4)
        if not generics:
            continue
        for tp in _type_params(generics):
            bounded = re.search(r"\b%s\s*:[^,>]*\bElementOps\b" % tp, generics) or re.search(r"\b%s\s*:[^,{]*\bElementOps\b" % tp, tail)
            assert bounded, "%s: type parameter %s has no ElementOps bound" % (name, tp)
            checked += 1
    assert checked >= 30
    sigs = {f["name"]: f for f in json.load(open(os.path.join(ROOT, "tools", "rust_shim", "signatures.json")))["functions"]}
    for name in ("gather", "where_op", "constant_of_shape"):
        assert sigs[name]["where"], name
        decl = txt[txt.index("pub fn %s<" % name):]
        decl = decl[:decl.index("{")]
        for clause in sigs[name]["where"].split(", "):
            assert clause.split(":")[0].strip() + ":" in decl and "AsI64" in decl or "AsI64" not in clause, (name, clause)


def _rust_items(txt):
    """(kind, owner, name, params, ret) of the public interface of a Rust source: free fns, methods by impl block, structs + pub fields"""
    owners = []
    for m in re.finditer(r"(?m)^impl(?:<[^>]*>)?\s+(?:(\w+)\s+for\s+)?(\w+)(?:<[^>]*>)?\s*\{", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"{": 1, "}": -1}.get(txt[i], 0)
            i += 1
        owners.append((m.start(), i, m.group(2), m.group(1) or ""))
    items = set()
    for m in re.finditer(r"(?m)^\s*pub fn (\w+)\s*(?:<[^>]*>)?\s*\(", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        params = re.sub(r"\s+", " ", txt[m.end():i - 1]).strip().rstrip(",")
        ret = re.sub(r"\s+", " ", txt[i:txt.index("{", i)]).strip()
        owner = next((o for s0, e0, o, tr in owners if s0 <= m.start() < e0 and not tr), "")
        items.add(("fn", owner, m.group(1), params, ret[2:].strip() if ret.startswith("->") else ""))
    for s0, e0, o, tr in owners:
        if tr in ("Default",):
            items.add(("impl", o, tr, "", ""))
    for m in re.finditer(r"(?m)^pub struct (\w+)(?:<[^>]*>)?\s*\{([^}]*)\}", txt):
        fields = tuple(re.sub(r"\s+", " ", f).strip() for f in re.findall(r"(?m)^\s*pub (\w+\s*:\s*.+?),?\s*$", m.group(2)))
        items.add(("struct", "", m.group(1), ", ".join(fields), ""))
    return items


def test_features_module_has_leles_public_interface():
    """lele::features (src/features/mod.rs:1-12): every public struct (with its public fields), free function, method (receiver
    included: `&self`, not `&mut self`) and Default impl the reference declares exists in the crate with the same declaration text
    -- extracted from both sides by the same rules (tools/rust_shim/extract_signatures.py wrote the reference's side into
    signatures.json; no reference checkout is needed here)."""
    want = json.load(open(os.path.join(ROOT, "tools", "rust_shim", "signatures.json")))["features"]
    have = _rust_items(open(os.path.join(CRATE, "features.rs")).read())
    missing = []
    for it in want:
        if it["kind"] == "fn":
            key = ("fn", it["owner"], it["name"], ", ".join(it["params"]), it["ret"])
        elif it["kind"] == "impl":
            key = ("impl", it["owner"], it["trait"], "", "")
        else:
            key = ("struct", "", it["name"], ", ".join(it["pub_fields"]), "")
        if key not in have:
            missing.append(key)
    assert not missing, missing
    assert len(want) >= 30


def test_the_sensevoice_example_is_expressible():
    """examples/sensevoice/src/main.rs:6,67-80 as text-level facts about the crate: `SenseVoiceFrontend::new(config)`; `compute` on an
    IMMUTABLE binding (`&self`) returning an owned view; `Cmvn::default()` and `cmvn.compute(&features)`; `.data.iter()` on a result
    (the payload derefs to a slice); generated model code's owned results do not leak a buffer per call."""
    feat = open(os.path.join(CRATE, "features.rs")).read()
    tens = open(os.path.join(CRATE, "tensor.rs")).read()
    rt = open(os.path.join(CRATE, "rt.rs")).read()
    assert "pub fn new(config: FeatureConfig) -> Self" in feat
    assert "pub fn compute(&self, pcm: &[f32]) -> TensorView<'static>" in feat
    assert "impl Default for Cmvn" in feat and "pub fn compute(&self, input: &TensorView) -> TensorView<'static>" in feat
    assert "&mut self" not in feat.replace("fn drop(&mut self)", "")
    assert re.search(r"impl<'a, T: ElementOps> Deref for Payload<'a, T>", tens) and "type Target = [T];" in tens
    assert "Box::leak" not in rt and "fn pooled_slot" in rt and "impl Drop for OwnedSlot" in rt
    # with_shape never re-labels a temporary copy as a declared-immutable weight (stale packed-weight cache hits)
    assert "Cow::Borrowed(b), weight } => Payload::Host { data: Cow::Borrowed(*b), weight: *weight }" in tens
    assert "Payload::Host { data: Cow::Owned(v.clone()), weight: false }" in tens


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
