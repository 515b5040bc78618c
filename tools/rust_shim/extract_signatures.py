#!/usr/bin/env python3
"""Build tools/rust_shim/signatures.json: the PUBLIC INTERFACE of lele's operator library -- for every `pub fn` the crate
re-exports as `lele::kernels::*` (src/kernels/mod.rs:23-39), `lele::tensor::TensorView` methods and `lele::features::*`, its
name, parameter list and return type, with the file:line it is declared at.  Runs only where the reference checkout is
mounted (/root/reference); its OUTPUT is committed, so that gen.py (and the tests) never need the reference.

Only declarations are read -- names, parameter names / types, return types: what a drop-in replacement has to reproduce
verbatim for generated model sources to link unchanged.  No function body is read or stored."""
import json
import os
import re
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
FILES = ["src/kernels/activations.rs", "src/kernels/conv1d.rs", "src/kernels/conv2d.rs", "src/kernels/fft.rs", "src/kernels/gemm.rs",
         "src/kernels/manipulation.rs", "src/kernels/math.rs", "src/kernels/norm.rs", "src/kernels/pooling.rs",
         "src/kernels/quantization.rs", "src/kernels/rnn.rs", "src/kernels/shape.rs", "src/kernels/utils.rs"]
# what mod.rs re-exports by name (the other modules are glob re-exports)
NAMED = {"conv1d.rs": {"conv1d", "conv1d_fused"},
         "conv2d.rs": {"conv_integer", "conv_integer_from_f32", "conv_integer_from_f32_multi", "conv_transpose", "conv2d", "conv2d_fused",
                       "conv2d_silu", "fused_scale_bias", "fused_scale_bias_silu", "gather_elements", "max_pool2d", "print_conv_stats", "reset_conv_stats",
                       "resize_nearest", "topk"},
         "gemm.rs": {"gemm", "matmul", "matmul_fused_add"}}
SKIP_MODULES = {"activations.rs", "fft.rs", "utils.rs"}  # not re-exported at lele::kernels (reachable as lele::kernels::<mod>::..)


def split_params(s):
    out, depth, cur = [], 0, ""
    for ch in s:
        if ch in "<([":
            depth += 1
        elif ch in ">)]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def main():
    sigs = []
    for rel in FILES:
        path = os.path.join(REF, rel)
        base = os.path.basename(rel)
        text = open(path).read()
        # strip #[cfg(test)] modules crudely: everything after "mod tests"
        cut = text.find("mod tests")
        if cut > 0:
            text = text[:cut]
        for m in re.finditer(r"(?m)^((?:#\[[^\]]*\]\s*)*)pub (unsafe )?fn (\w+)\s*(<[^>]*(?:<[^>]*>[^>]*)*>)?\s*\(", text):
            attrs, unsafe, name, generics = m.group(1) or "", m.group(2), m.group(3), m.group(4) or ""
            if "target_arch" in attrs and "x86_64" not in attrs:
                continue
            # parameter list: up to the matching ')'
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            params = split_params(re.sub(r"\s+", " ", text[m.end():i - 1]))
            rest = text[i:text.find("{", i)]
            ret = re.sub(r"\s+", " ", rest.split("where")[0]).strip()
            ret = ret[2:].strip() if ret.startswith("->") else ""
            where = re.sub(r"\s+", " ", rest.split("where", 1)[1]).strip().rstrip(",") if "where" in rest else ""
            line = text.count("\n", 0, m.start()) + 1
            exported = base not in SKIP_MODULES and (base not in NAMED or name in NAMED[base])
            sigs.append({"name": name, "generics": re.sub(r"\s+", " ", generics), "params": [p for p in params if p], "ret": ret, "where": where,
                         "unsafe": bool(unsafe), "at": "%s:%d" % (rel, line), "exported": exported})
    seen, uniq = set(), []
    for s in sorted(sigs, key=lambda s: not s["exported"]):  # cfg-duplicated functions: keep the first; re-exported ones win a name clash
        if s["name"] in seen:
            continue
        seen.add(s["name"])
        uniq.append(s)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "signatures.json")
    feats = features_interface()
    tens = tensor_interface()
    json.dump({"source": "miuda-ai/lele src/kernels + src/features + src/tensor.rs (declarations only)", "functions": uniq, "features": feats, "tensor": tens},
              open(out, "w"), indent=1)
    print(len(uniq), "signatures ->", out, "; exported:", sum(1 for s in uniq if s["exported"]), "; features items:", len(feats), "; tensor items:", len(tens))
    write_api_tokens()


def features_interface():
    """`lele::features::*` (src/features/mod.rs:1-12 re-exports every module): public structs with their public fields, public free
    functions and public methods (with the impl block's type as `owner`), declarations only"""
    items = []
    for base in ("window.rs", "mel.rs", "fft.rs", "lfr.rs", "cmvn.rs", "pipeline.rs"):
        rel = "src/features/" + base
        text = open(os.path.join(REF, rel)).read()
        cut = text.find("mod tests")
        if cut > 0:
            text = text[:cut]
        # impl blocks: owner by brace matching
        owners = []  # (start, end, owner, trait)
        for m in re.finditer(r"(?m)^impl(?:<[^>]*>)?\s+(?:(\w+)\s+for\s+)?(\w+)(?:<[^>]*>)?\s*\{", text):
            i, depth = m.end(), 1
            while depth:
                depth += {"{": 1, "}": -1}.get(text[i], 0)
                i += 1
            owners.append((m.start(), i, m.group(2), m.group(1) or ""))
        for m in re.finditer(r"(?m)^(\s*)pub fn (\w+)\s*(<[^>]*>)?\s*\(", text):
            i, depth = m.end(), 1
            while depth:
                depth += {"(": 1, ")": -1}.get(text[i], 0)
                i += 1
            params = split_params(re.sub(r"\s+", " ", text[m.end():i - 1]))
            rest = text[i:text.find("{", i)]
            ret = re.sub(r"\s+", " ", rest.split("where")[0]).strip()
            ret = ret[2:].strip() if ret.startswith("->") else ""
            owner = next((o for s0, e0, o, tr in owners if s0 <= m.start() < e0 and not tr), "")
            items.append({"kind": "fn", "owner": owner, "name": m.group(2), "generics": m.group(3) or "", "params": [p for p in params if p], "ret": ret,
                          "at": "%s:%d" % (rel, text.count("\n", 0, m.start()) + 1)})
        for s0, e0, o, tr in owners:
            if tr:
                items.append({"kind": "impl", "owner": o, "trait": tr, "at": "%s:%d" % (rel, text.count("\n", 0, s0) + 1)})
        for m in re.finditer(r"(?m)^pub struct (\w+)(<[^>]*>)?\s*\{([^}]*)\}", text):
            fields = [re.sub(r"\s+", " ", f).strip() for f in re.findall(r"(?m)^\s*pub (\w+\s*:\s*.+?),?\s*$", m.group(3))]
            items.append({"kind": "struct", "name": m.group(1), "generics": m.group(2) or "", "pub_fields": fields,
                          "at": "%s:%d" % (rel, text.count("\n", 0, m.start()) + 1)})
    return items


def tensor_interface():
    """`lele::tensor` (src/tensor.rs): every public item's DECLARATION -- methods (with the impl block's self type, `unsafe`, generics,
    parameters, return type), type aliases, traits with their method declarations, trait impls (trait, generics, for-type), re-exports"""
    rel = "src/tensor.rs"
    text = open(os.path.join(REF, rel)).read()
    line = lambda pos: text.count("\n", 0, pos) + 1   # noqa: E731
    items = []
    for m in re.finditer(r"(?m)^pub use ([^;]+);", text):
        items.append({"kind": "use", "path": re.sub(r"\s+", " ", m.group(1)), "at": "%s:%d" % (rel, line(m.start()))})
    for m in re.finditer(r"(?m)^pub type (\w+)(<[^>]*>)?\s*=\s*([^;]+);", text):
        items.append({"kind": "type", "name": m.group(1), "generics": m.group(2) or "", "target": re.sub(r"\s+", " ", m.group(3)), "at": "%s:%d" % (rel, line(m.start()))})
    blocks = []   # (start, end, header text)
    for m in re.finditer(r"(?m)^(impl|pub trait)\b([^{]*)\{", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"{": 1, "}": -1}.get(text[i], 0)
            i += 1
        blocks.append((m.start(), i, m.group(1), re.sub(r"\s+", " ", m.group(2)).strip()))
    for s0, e0, kw, head in blocks:
        if kw == "pub trait":
            name = re.match(r"(\w+)", head).group(1)
            items.append({"kind": "trait", "name": name, "header": head.split(" where")[0].strip(), "at": "%s:%d" % (rel, line(s0))})
        else:
            mm = re.match(r"(<.*?>)?\s*(?:(\w+)(<[^>]*>)?\s+for\s+)?(.+?)(?:\s+where\b.*)?$", head)
            if mm.group(2):
                items.append({"kind": "impl", "trait": mm.group(2) + (mm.group(3) or ""), "generics": mm.group(1) or "", "for": mm.group(4).strip(),
                              "at": "%s:%d" % (rel, line(s0))})
    for m in re.finditer(r"(?m)^\s*(pub )?(unsafe )?fn (\w+)\s*(<[^>(]*>)?\s*\(", text):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(text[i], 0)
            i += 1
        params = split_params(re.sub(r"\s+", " ", text[m.end():i - 1]))
        j = min(x for x in (text.find("{", i), text.find(";", i)) if x >= 0)
        ret = re.sub(r"\s+", " ", text[i:j].split("where")[0]).strip()
        ret = ret[2:].strip() if ret.startswith("->") else ""
        blk = next(((kw, head) for s0, e0, kw, head in blocks if s0 <= m.start() < e0), None)
        if blk is None or (not m.group(1) and blk[0] != "pub trait"):
            continue   # private helpers; trait-impl bodies are covered by the "impl" items
        owner = blk[1].split(" where")[0].strip()
        items.append({"kind": "fn", "owner": owner, "name": m.group(3), "unsafe": bool(m.group(2)), "generics": m.group(4) or "",
                      "params": [p for p in params if p], "ret": ret, "at": "%s:%d" % (rel, line(m.start()))})
    return items


def write_api_tokens():
    """Every `TensorView::<ident>` / `lele::<path>` token, `.method(` called on a view and trait name that (a) the reference's one
    generated model source and (b) the text its compiler emits (src/compiler/**: the helper block mod.rs:1135-1233, the per-operator
    emitters, snippets) and (c) the example applications contain -- the full surface a drop-in `lele` crate has to resolve.  Stored under
    "api_tokens" of tests/golden/generated_kernel_names.json next to the kernel-name lists (names only: data for tests/test_rust_shim.py)."""
    pat = re.compile(r"(?:lele::)?(?:tensor::)?TensorView(?:F32|I8|U8|I32|I64|F16|BF16)?::(?:<[^>]*>::)?[a-z_0-9]+|lele::[a-zA-Z_0-9]+(?:::[a-zA-Z_0-9]+)*|\bIntoLogits\b|\binto_logits\b"
                     r"|TensorView(?:F32|I8|U8|I32|I64|F16|BF16)\b")
    groups = {"yolo26seg": ["examples/yolo26n-seg/src/yolo26seg.rs"], "emitter": [], "apps": []}
    for root, _d, files in os.walk(os.path.join(REF, "src", "compiler")):
        groups["emitter"] += [os.path.relpath(os.path.join(root, f), REF) for f in files if f.endswith(".rs")]
    for root, _d, files in os.walk(os.path.join(REF, "examples")):
        for f in files:
            rp = os.path.relpath(os.path.join(root, f), REF)
            if f.endswith(".rs") and rp not in groups["yolo26seg"] and "/target/" not in rp:
                groups["apps"].append(rp)
    out = {}
    for g, paths in groups.items():
        toks = set()
        for rp in sorted(paths):
            txt = open(os.path.join(REF, rp)).read()
            txt = re.sub(r"(?m)^\s*//.*$", "", txt)
            found = pat.findall(txt)
            for m in re.finditer(r"\buse (lele::[a-z_:]+)::\{([^}]*)\}", txt):   # `use lele::features::{A, B}` -> lele::features::A, ..
                found += ["%s::%s" % (m.group(1), n.strip()) for n in m.group(2).split(",") if n.strip()]
            for t in found:
                t = re.sub(r"::<[^>]*>", "", t)
                if t.startswith("TensorView::"):
                    t = "lele::tensor::" + t
                elif t.startswith("tensor::TensorView::"):
                    t = "lele::" + t
                elif re.match(r"TensorView[A-Z0-9]+$", t) or t in ("IntoLogits",):
                    t = "lele::tensor::" + t
                elif t == "into_logits":
                    t = "lele::tensor::IntoLogits::into_logits"
                if t.rstrip(":") in ("lele", "lele::kernels", "lele::tensor", "lele::features", "lele::compiler", "lele::model"):
                    continue
                if t.startswith("lele::compiler") or t.startswith("lele::model"):
                    continue   # the ONNX compiler itself (feature "compiler"): out of scope, never referenced by generated sources
                toks.add(t)
        out[g] = sorted(toks)
    gp = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "generated_kernel_names.json")
    d = json.load(open(gp))
    d["api_tokens"] = out
    d["api_tokens_note"] = ("every lele::<path> / TensorView::<fn> / IntoLogits token in (yolo26seg) the reference's generated model source, (emitter) the text "
                            "src/compiler/** emits, (apps) examples/**.rs -- tools/rust_shim/extract_signatures.py; tests/test_rust_shim.py resolves each in rust/lele-hip")
    json.dump(d, open(gp, "w"), indent=1)
    print("api tokens:", {k: len(v) for k, v in out.items()}, "->", gp)


if __name__ == "__main__":
    main()
