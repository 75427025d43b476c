#!/usr/bin/env python3
"""Generate bindings/sys.rs -- the complete `extern "C"` mirror of include/astroburst_hip.h for the Rust host
(src-tauri), one `#[repr(C)]` struct per C struct and one declaration per AB_API entry point.

    python tools/gen_rust_sys.py            # rewrite bindings/sys.rs
    python tools/gen_rust_sys.py --check    # fail if the committed file is stale

rustc is not available in this image, so the file cannot be compiled here; what IS checked (tests/test_abi_cpu.py) is that
every struct's size and field offsets under Rust's #[repr(C)] rules (natural alignment, declaration order -- computed by
`layout()` below from the generated Rust types) equal what gcc reports for the C header, and that every exported symbol of
the header appears exactly once.
"""
import argparse
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "astroburst_hip.h")
OUT = os.path.join(ROOT, "bindings", "sys.rs")

PRIM = {   # C type -> (Rust type, size, align)
    "float": ("f32", 4, 4), "double": ("f64", 8, 8), "int": ("c_int", 4, 4), "int32_t": ("i32", 4, 4), "uint32_t": ("u32", 4, 4),
    "int64_t": ("i64", 8, 8), "uint64_t": ("u64", 8, 8), "size_t": ("usize", 8, 8), "uint8_t": ("u8", 1, 1), "char": ("c_char", 1, 1),
    "void": ("c_void", 0, 1), "unsigned": ("c_uint", 4, 4),
}
OPAQUE = ("ab_ctx", "ab_comm")


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def parse(text):
    """-> (structs {name: [(field, ctype, is_ptr, array_len)]}, enums {name: [(ident, value)]}, funcs [(ret, name, [(ctype, is_ptr, pname, arr)])],
    defines {name: value})"""
    src = strip_comments(text)
    structs, enums, funcs = {}, {}, []
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            decl = decl.replace("const ", "")
            mm = re.match(r"(\w+)\s+(.*)$", decl)
            ctype, rest = mm.group(1), mm.group(2)
            for item in rest.split(","):
                item = item.strip()
                ptr = item.count("*")
                item = item.replace("*", "").strip()
                dims = [int(x) for x in re.findall(r"\[(\w+)\]", item) if x.isdigit()]
                name = re.sub(r"\[.*", "", item)
                fields.append((name, ctype, ptr, dims))
        structs[m.group(2)] = fields
    for m in re.finditer(r"typedef\s+enum\s*\{(.*?)\}\s*(\w+)\s*;", src, re.S):
        vals, nxt = [], 0
        for item in m.group(1).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                k, v = [x.strip() for x in item.split("=")]
                nxt = int(v, 0)
            else:
                k = item
            vals.append((k, nxt))
            nxt += 1
        enums[m.group(2)] = vals
    for m in re.finditer(r"AB_API\s+([\w\s\*]+?)\b(ab_\w+)\s*\((.*?)\)\s*;", src, re.S):
        ret = " ".join(m.group(1).split())
        params = []
        ptext = " ".join(m.group(3).split())
        if ptext and ptext != "void":
            for i, p in enumerate(ptext.split(",")):
                p = p.strip()
                if "(*" in p:  # function pointer typedef'd elsewhere is passed by name; none inline in this header
                    raise ValueError(p)
                const = "const " in p
                p2 = p.replace("const ", "")
                arr = re.findall(r"\[(\w*)\]", p2)
                p2 = re.sub(r"\[\w*\]", "", p2)
                ptr = p2.count("*") + (1 if arr else 0)
                toks = p2.replace("*", " ").split()
                ctype, pname = toks[0], (toks[1] if len(toks) > 1 else f"arg{i}")
                params.append((ctype, ptr, pname, const))
        funcs.append((ret, m.group(2), params))
    defines = dict(re.findall(r"#define\s+(AB_[A-Z_]+)\s+(\d+)", src))
    cb = re.search(r"typedef\s+void\s*\(\*(\w+)\)\s*\((.*?)\)\s*;", src, re.S)
    return structs, enums, funcs, defines, cb


def rust_type(ctype, ptr, const, structs, enums):
    if ctype in PRIM:
        base = PRIM[ctype][0]
    elif ctype in structs or ctype in OPAQUE:
        base = ctype
    elif ctype in enums:
        base = "c_int"
    elif ctype == "ab_progress_cb":
        base = "ab_progress_cb"
    else:
        raise KeyError(ctype)
    for _ in range(ptr):
        base = ("*const " if const else "*mut ") + base
    return base


def layout(structs, name, _memo={}):
    """(size, align, [(field, offset, size)]) of struct `name` under #[repr(C)] == the C ABI's rules"""
    if name in _memo:
        return _memo[name]
    off, align, out = 0, 1, []
    for fname, ctype, ptr, dims in structs[name]:
        if ptr:
            sz, al = 8, 8
        elif ctype in PRIM:
            _, sz, al = PRIM[ctype]
        else:
            sz, al, _ = layout(structs, ctype)
        n = 1
        for d in dims:
            n *= d
        off = (off + al - 1) // al * al
        out.append((fname, off, sz * n))
        off += sz * n
        align = max(align, al)
    size = (off + align - 1) // align * align
    _memo[name] = (size, align, out)
    return _memo[name]


def generate():
    text = open(HEADER).read()
    structs, enums, funcs, defines, cb = parse(text)
    L = []
    L.append("// GENERATED by tools/gen_rust_sys.py from include/astroburst_hip.h -- do not edit.")
    L.append("// The raw FFI layer of the AstroBurst HIP core for the Rust host (src-tauri): `mod sys` of INTEGRATION.md.")
    L.append("// Every entry point returns an ab_status (0 = AB_OK) and never unwinds; see the header for the semantics and the")
    L.append("// reference function (core::*) each one replaces.")
    L.append("#![allow(non_camel_case_types, dead_code)]")
    L.append("use std::os::raw::{c_char, c_int, c_uint, c_void};")
    L.append("")
    for name, val in defines.items():
        L.append(f"pub const {name}: usize = {val};")
    L.append("")
    for name, vals in enums.items():
        L.append(f"// enum {name}")
        for k, v in vals:
            L.append(f"pub const {k}: c_int = {v};")
        L.append("")
    for o in OPAQUE:
        L.append("#[repr(C)]")
        L.append(f"pub struct {o} {{ _private: [u8; 0] }}")
    L.append("")
    if cb:
        L.append(f"pub type {cb.group(1)} = Option<unsafe extern \"C\" fn(stage: *const c_char, current: u64, total: u64, user: *mut c_void)>;")
        L.append("")
    for name, fields in structs.items():
        size, align, lay = layout(structs, name)
        L.append(f"/// size {size}, align {align}")
        L.append("#[repr(C)]")
        L.append("#[derive(Clone, Copy, Debug)]")
        L.append(f"pub struct {name} {{")
        for (fname, ctype, ptr, dims), (_, off, _) in zip(fields, lay):
            t = rust_type(ctype, ptr, ctype == "float" and name == "ab_plane" or ctype == "char" or (name in ("ab_calibration_masters", "ab_batch_channel_input") and ctype == "ab_plane"),
                          structs, enums)
            for d in reversed(dims):
                t = f"[{t}; {d}]"
            L.append(f"    pub {fname}: {t}, // offset {off}")
        L.append("}")
        L.append("")
    L.append('#[link(name = "astroburst_hip")]')
    L.append('extern "C" {')
    for ret, name, params in funcs:
        ps = []
        for ctype, ptr, pname, const in params:
            if pname in ("type", "ref", "in", "fn", "box", "loop", "match", "move", "mod", "impl"):
                pname += "_"
            ps.append(f"{pname}: {rust_type(ctype, ptr, const, structs, enums)}")
        rt = ret.replace("const ", "").strip()
        if rt == "void":
            r = ""
        elif rt.endswith("*"):
            base = rt[:-1].strip()
            r = " -> " + ("*const " if "const" in ret else "*mut ") + (PRIM[base][0] if base in PRIM else base)
        else:
            r = " -> " + (PRIM[rt][0] if rt in PRIM else rt)
        L.append(f"    pub fn {name}({', '.join(ps)}){r};")
    L.append("}")
    L.append("")
    return "\n".join(L), structs, funcs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    text, structs, funcs = generate()
    if args.check:
        old = open(OUT).read() if os.path.exists(OUT) else ""
        if old != text:
            print("bindings/sys.rs is stale: run python tools/gen_rust_sys.py")
            sys.exit(1)
        print(f"bindings/sys.rs is current: {len(structs)} structs, {len(funcs)} functions")
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    open(OUT, "w").write(text)
    print(f"wrote {OUT}: {len(structs)} structs, {len(funcs)} functions")


if __name__ == "__main__":
    main()
