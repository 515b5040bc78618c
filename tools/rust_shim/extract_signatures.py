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
    json.dump({"source": "miuda-ai/lele src/kernels + src/features (declarations only)", "functions": uniq, "features": feats}, open(out, "w"), indent=1)
    print(len(uniq), "signatures ->", out, "; exported:", sum(1 for s in uniq if s["exported"]), "; features items:", len(feats))


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


if __name__ == "__main__":
    main()
