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
    assert re.search(r"impl<'a, T: Clone> Deref for Payload<'a, T>", tens) and "type Target = [T];" in tens   # no bound beyond upstream's
    assert "Box::leak" not in rt and "fn pooled_slot" in rt and "impl Drop for OwnedSlot" in rt
    # with_shape never re-labels a temporary copy as a declared-immutable weight (stale packed-weight cache hits)
    assert "Cow::Borrowed(b), weight } => Payload::Host { data: Cow::Borrowed(*b), weight: *weight }" in tens
    assert "Payload::Host { data: Cow::Owned(v.clone()), weight: false }" in tens


def _tensor_items(txt):
    """the public interface of a `tensor.rs`, by the rules tools/rust_shim/extract_signatures.py::tensor_interface applied to the reference's"""
    items = []
    for m in re.finditer(r"(?m)^pub use ([^;]+);", txt):
        items.append(("use", re.sub(r"\s+", " ", m.group(1))))
    for m in re.finditer(r"(?m)^pub type (\w+)(<[^>]*>)?\s*=\s*([^;]+);", txt):
        items.append(("type", m.group(1), m.group(2) or "", re.sub(r"\s+", " ", m.group(3))))
    blocks = []
    for m in re.finditer(r"(?m)^(impl|pub trait)\b([^{]*)\{", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"{": 1, "}": -1}.get(txt[i], 0)
            i += 1
        blocks.append((m.start(), i, m.group(1), re.sub(r"\s+", " ", m.group(2)).strip()))
    for s0, e0, kw, head in blocks:
        if kw == "pub trait":
            items.append(("trait", head.split(" where")[0].strip()))
        else:
            mm = re.match(r"(<.*?>)?\s*(?:(\w+)(<[^>]*>)?\s+for\s+)?(.+?)(?:\s+where\b.*)?$", head)
            if mm.group(2):
                items.append(("impl", mm.group(2) + (mm.group(3) or ""), mm.group(1) or "", mm.group(4).strip()))
    for m in re.finditer(r"(?m)^\s*(pub )?(unsafe )?fn (\w+)\s*(<[^>(]*>)?\s*\(", txt):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(txt[i], 0)
            i += 1
        params = re.sub(r"\s+", " ", txt[m.end():i - 1]).strip().rstrip(",")
        j = min(x for x in (txt.find("{", i), txt.find(";", i)) if x >= 0)
        ret = re.sub(r"\s+", " ", txt[i:j].split("where")[0]).strip()
        blk = next(((kw, head) for s0, e0, kw, head in blocks if s0 <= m.start() < e0), None)
        if blk is None or (not m.group(1) and blk[0] != "pub trait"):
            continue
        items.append(("fn", blk[1].split(" where")[0].strip(), m.group(3), bool(m.group(2)), m.group(4) or "", params, ret[2:].strip() if ret.startswith("->") else ""))
    return items


def test_tensor_module_has_leles_public_interface():
    """lele::tensor (src/tensor.rs): every public item the reference declares -- the half re-export, the seven TensorView* aliases, every
    TensorView method (self type of its impl block, `unsafe`, generics, parameter list, return type: `new(&'a [T], &'a [usize])`,
    `unsafe fn detach<'b>`, the nine `from_bytes_*` decoders ...), the IntoLogits trait, its method and its two impls -- exists in the crate
    with the same declaration text.  A generated model's helper block (src/compiler/mod.rs:1135-1233) and examples/sensevoice/src/main.rs:7
    are written against exactly these."""
    want = json.load(open(os.path.join(ROOT, "tools", "rust_shim", "signatures.json")))["tensor"]
    have = set(_tensor_items(open(os.path.join(CRATE, "tensor.rs")).read()))
    missing = []
    for it in want:
        if it["kind"] == "use":
            key = ("use", it["path"])
        elif it["kind"] == "type":
            key = ("type", it["name"], it["generics"], it["target"])
        elif it["kind"] == "trait":
            key = ("trait", it["header"])
        elif it["kind"] == "impl":
            key = ("impl", it["trait"], it["generics"], it["for"])
        else:
            key = ("fn", it["owner"], it["name"], it["unsafe"], it["generics"], ", ".join(it["params"]), it["ret"])
        if key not in have:
            missing.append(key)
    assert not missing, missing
    assert len(want) >= 30
    # the struct keeps upstream's two public fields, by name
    tens = open(os.path.join(CRATE, "tensor.rs")).read()
    body = re.search(r"pub struct TensorView<'a, T[^>]*>\s*\{([^}]*)\}", tens).group(1)
    assert re.findall(r"pub (\w+):", body) == ["data", "shape"]
    assert "#[derive(Debug, Clone)]\npub struct TensorView" in tens      # yolo26seg.rs:653 clones a view; tensor.rs:4


def _crate_paths():
    """every path a `lele::...` token can resolve to in the crate: module -> set of public names (functions, structs, traits, aliases,
    re-exports), plus TensorView's associated functions and IntoLogits' method"""
    src = {n: open(os.path.join(CRATE, n + ".rs")).read() for n in ("kernels", "tensor", "features", "rt", "lib")}
    def names(txt):   # noqa: E306
        out = set(re.findall(r"(?m)^\s*pub (?:unsafe )?fn (\w+)", txt)) | set(re.findall(r"(?m)^pub (?:struct|trait|enum|type) (\w+)", txt))
        out |= set(re.findall(r"(?m)^\s*pub use [\w:]+ as (\w+);", txt))
        for m in re.finditer(r"(?m)^\s*pub use ([\w:]+)::\{([^}]*)\};", txt):
            out |= {n.strip().split(" as ")[-1] for n in m.group(2).split(",")}
        out |= {m.split("::")[-1] for m in re.findall(r"(?m)^\s*pub use ([\w:]+);", txt)}
        return out
    paths = {"lele::kernels": names(src["kernels"]), "lele::tensor": names(src["tensor"]), "lele::features": names(src["features"])}
    for m in re.finditer(r"(?m)^pub mod (\w+) \{", src["kernels"]):
        i, depth = m.end(), 1
        while depth:
            depth += {"{": 1, "}": -1}.get(src["kernels"][i], 0)
            i += 1
        paths["lele::kernels::" + m.group(1)] = names(src["kernels"][m.end():i])
    if "pub use kernels::*;" in src["lib"]:
        paths["lele"] = set(paths["lele::kernels"]) | {"kernels", "tensor", "features"}
    paths["lele::tensor::TensorView"] = {it[2] for it in _tensor_items(src["tensor"]) if it[0] == "fn" and "TensorView<" in it[1]}
    paths["lele::tensor::IntoLogits"] = {it[2] for it in _tensor_items(src["tensor"]) if it[0] == "fn" and it[1].startswith("IntoLogits")}
    return paths


def test_every_api_token_of_generated_and_example_sources_resolves():
    """not only kernel NAMES.  Every `lele::<path>` / `TensorView::<fn>` / IntoLogits token of (a) the reference's generated
    Yolo26n-seg source, (b) the text the reference's compiler emits and (c) its example applications (tests/golden/generated_kernel_names.json
    "api_tokens", written by tools/rust_shim/extract_signatures.py) names an item this crate defines at that path."""
    toks = json.load(open(os.path.join(ROOT, "tests", "golden", "generated_kernel_names.json")))["api_tokens"]
    paths = _crate_paths()
    not_defined_upstream = {"lele::kernels::argmax"}   # written by src/compiler/ops but no such fn exists in src/kernels (would not compile upstream either)
    unresolved = []
    for group, lst in toks.items():
        for t in lst:
            if t in not_defined_upstream:
                continue
            mod, _, name = t.rpartition("::")
            if not (mod in paths and name in paths[mod]):
                unresolved.append((group, t))
    assert not unresolved, unresolved
    assert sum(len(v) for v in toks.values()) >= 150
    for must in ("lele::tensor::TensorView::from_bytes_f32", "lele::tensor::TensorView::from_bytes_i32_as_i64", "lele::kernels::utils::cast_to_i64",
                 "lele::kernels::timing::reset", "lele::tensor::IntoLogits", "lele::tensor::TensorView::new"):
        assert any(must in lst for lst in toks.values()), must


def test_weight_constructors_mark_the_payload_as_a_declared_weight():
    """`weight_f32` -> `from_bytes_f32` is where a weights.bin slice becomes LELE_MEM_WEIGHT (uploaded / packed once, cached by pointer):
    the aligned path goes through `TensorView::weight`, the u8 / i8 / f16 decoders keep ONE decoded image per slice at a stable address."""
    tens = open(os.path.join(CRATE, "tensor.rs")).read()
    f32_body = tens[tens.index("pub fn from_bytes_f32"):tens.index("pub fn from_bytes_u8")]
    assert "TensorView::weight(words, shape)" in f32_body and "from_owned" in f32_body      # aligned: in place + weight; unaligned: owned copy
    for name, kind in (("from_bytes_u8", 0), ("from_bytes_i8", 1), ("from_bytes_f16", 2)):
        body = tens[tens.index("pub fn %s" % name):]
        assert "decoded_weight(bytes, %d, shape" % kind in body[:body.index("\n    }")], name
    dw = tens[tens.index("fn decoded_weight"):tens.index("impl<'a> TensorView<'a, f32> {\n    /// tensor.rs:131")]
    assert "weight: true" in dw and "into_boxed_slice" in dw and "(bytes.as_ptr() as usize, bytes.len(), kind)" in dw
    assert "if *weight { ffi::LELE_MEM_WEIGHT } else { ffi::LELE_MEM_HOST }" in tens


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
