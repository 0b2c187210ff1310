"""The Rust shim of INTEGRATION.md cannot be compiled here (no Rust toolchain in the image), so nothing but this test
keeps it from drifting: every `#[repr(C)] struct` in the document must have the fields of its C twin in
include/gfasort_hip.h — same names, same order, same widths — and every `fn gfs_*` in its `extern "C"` blocks must be
declared in the header with the same number of parameters and matching parameter kinds."""
import os
import re

from util import ROOT

RUST_TO_C = {"u64": "uint64_t", "u32": "uint32_t", "i32": "int32_t", "u8": "uint8_t", "f64": "double", "c_int": "int"}
STRUCT_TWINS = {"GfsGraphView": "gfs_graph_view", "GfsSgdParams": "gfs_sgd_params", "GfsLayoutParams": "gfs_layout_params",
                "GfsLaunchConfig": "gfs_launch_config", "GfsStats": "gfs_stats", "GfsRankConfig": "gfs_rank_config",
                "GfsRankInfo": "gfs_rank_info"}


def _strip_c_comments(t):
    return re.sub(r"/\*.*?\*/", "", t, flags=re.S)


def _c_structs(text):
    out = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            mm = re.match(r"(const\s+)?(\w+)\s*(\*?)\s*([\w\s,\[\]]+)$", decl)      # all field types are one word
            ctype = ("const " if mm.group(1) else "") + mm.group(2).strip() + ("*" if mm.group(3) else "")
            for name in mm.group(4).split(","):
                name = name.strip()
                arr = re.match(r"(\w+)\[(\d+)\]", name)
                fields.append((arr.group(1), f"{ctype}[{arr.group(2)}]") if arr else (name, ctype))
        out[m.group(3)] = fields
    return out


def _rust_structs(text):
    out = {}
    for m in re.finditer(r"pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        if m.group(1) not in STRUCT_TWINS:
            continue
        body = re.sub(r"//[^\n]*", "", m.group(2))
        fields = []
        for f in body.split(","):
            f = f.strip()
            if not f:
                continue
            mm = re.match(r"pub (\w+)\s*:\s*(.+)$", f, flags=re.S)
            fields.append((mm.group(1), mm.group(2).strip()))
        out[m.group(1)] = fields
    return out


def _rust_type_to_c(t):
    t = t.strip()
    arr = re.match(r"\[(\w+);\s*(\d+)\]", t)
    if arr:
        return f"{RUST_TO_C[arr.group(1)]}[{arr.group(2)}]"
    if t.startswith("*const "):
        return "const " + _rust_type_to_c(t[7:]) + "*"
    if t.startswith("*mut "):
        return _rust_type_to_c(t[5:]) + "*"
    if t in STRUCT_TWINS:
        return STRUCT_TWINS[t]
    return RUST_TO_C.get(t, t)


def _docs():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as fh:
        doc = fh.read()
    with open(os.path.join(ROOT, "include", "gfasort_hip.h")) as fh:
        hdr = _strip_c_comments(fh.read())
    rust = "\n".join(re.findall(r"```rust\n(.*?)```", doc, flags=re.S))
    return rust, hdr


def test_rust_structs_mirror_the_header_field_for_field():
    rust, hdr = _docs()
    rs, cs = _rust_structs(rust), _c_structs(hdr)
    for rname, cname in STRUCT_TWINS.items():
        assert rname in rs, f"INTEGRATION.md lacks {rname}"
        assert cname in cs, f"header lacks {cname}"
        got = [(n, _rust_type_to_c(t)) for n, t in rs[rname]]
        want = [(n, t.replace(" *", "*")) for n, t in cs[cname]]
        assert got == want, (rname, got, want)


def _c_functions(hdr):
    out = {}
    for m in re.finditer(r"\b([\w\s\*]+?)\b(gfs_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S):
        args = [a.strip() for a in m.group(3).replace("\n", " ").split(",")]
        out[m.group(2)] = [] if args == ["void"] else args
    return out


def test_rust_extern_functions_exist_in_the_header_with_matching_parameters():
    rust, hdr = _docs()
    cfn = _c_functions(hdr)
    seen = 0
    for m in re.finditer(r"\bfn (gfs_\w+)\s*\((.*?)\)\s*(?:->\s*([\w\s\*]+))?;", rust, flags=re.S):
        name, params = m.group(1), m.group(2)
        params = re.sub(r"//[^\n]*", "", params)
        rparams = [p.strip() for p in params.split(",") if p.strip()]
        assert name in cfn, f"INTEGRATION.md binds {name}, which the header does not declare"
        cparams = cfn[name]
        assert len(rparams) == len(cparams), (name, rparams, cparams)
        for rp, cp in zip(rparams, cparams):
            rtype = rp.split(":", 1)[1].strip()
            is_ptr_r = rtype.startswith("*") or rtype == "GfsAllreduceFn"
            is_ptr_c = "*" in cp or "gfs_allreduce_fn" in cp
            assert is_ptr_r == is_ptr_c, (name, rp, cp)
            if not is_ptr_r:
                ctype = " ".join(cp.split()[:-1])
                assert _rust_type_to_c(rtype) == ctype, (name, rp, cp)
            elif rtype.startswith("*const") and "c_void" not in rtype and "c_char" not in rtype:
                assert cp.startswith("const "), (name, rp, cp)
        seen += 1
    assert seen >= 12
