#!/usr/bin/env python3
"""Generates the Rust FFI block of INTEGRATION.md (section 2) from include/blubhip.h, so that the binding a maintainer would paste
into `src/simulation/blubhip_sys.rs` always covers the whole header.  `python tools/gen_rust_ffi.py --write` rewrites the block
between the GENERATED markers of INTEGRATION.md; tests/test_host_abi.py checks that the committed block is up to date."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "blubhip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED FFI (tools/gen_rust_ffi.py) -->", "<!-- END GENERATED FFI -->"

PRIM = {"int": "c_int", "unsigned": "u32", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "int64_t": "i64", "uint8_t": "u8", "int8_t": "i8",
        "float": "f32", "double": "f64", "size_t": "usize", "char": "c_char", "void": "c_void"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    return re.sub(r"//[^\n]*", " ", text)


def rust_type(ctype, array=""):
    """'const float*' -> '*const f32'; array = '[3]' / '[N][3]' (only used for struct fields)."""
    t = ctype.strip()
    const = False
    stars = t.count("*")
    t = t.replace("*", " ").split()
    if "const" in t:
        const = True
        t = [x for x in t if x != "const"]
    t = [x for x in t if x not in ("struct", "enum")]
    base = PRIM.get(t[-1], t[-1])
    for _ in range(stars):
        base = ("*const " if const else "*mut ") + base
        const = False if stars > 1 else const
    if array:
        dims = re.findall(r"\[([^\]]+)\]", array)
        for d in reversed(dims):
            base = "[%s; %s as usize]" % (base, d) if not d.isdigit() else "[%s; %s]" % (base, d)
    return base


def parse(header_text):
    src = strip_comments(header_text)
    consts, structs, funcs, opaque = [], [], [], []
    for m in re.finditer(r"enum\s*(\w*)\s*\{([^}]*)\}\s*(\w*)\s*;", src):
        val = -1
        for item in m.group(2).split(","):
            item = item.strip()
            if not item:
                continue
            if "=" in item:
                name, v = [x.strip() for x in item.split("=")]
                val = int(v, 0)
            else:
                name, val = item, val + 1
            consts.append((name, val))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", src):
        opaque.append(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", src, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            fp = re.match(r"(.+?)\(\s*\*\s*(\w+)\s*\)\s*\((.*)\)$", decl)
            if fp:   # function pointer field
                args = ", ".join(rust_type(re.sub(r"\b\w+$", "", a.strip()) if re.search(r"[\w\*]\s+\w+$", a.strip()) else a) for a in fp.group(3).split(","))
                ret = rust_type(fp.group(1))
                fields.append((fp.group(2), "Option<unsafe extern \"C\" fn(%s)%s>" % (args, "" if ret == "c_void" else " -> " + ret)))
                continue
            mm = re.match(r"(.+?)\s*(\w+(?:\s*\[[^\]]+\])*(?:\s*,\s*\w+(?:\s*\[[^\]]+\])*)*)$", decl)
            ctype, names = mm.group(1), mm.group(2)
            while ctype.endswith("*"):
                pass_star = True
                break
            for nm in names.split(","):
                nm = nm.strip()
                arr = "".join(re.findall(r"\[[^\]]+\]", nm))
                nm = re.sub(r"\[[^\]]+\]", "", nm).strip()
                stars = ""
                while nm.startswith("*"):
                    stars, nm = stars + "*", nm[1:]
                fields.append((nm, rust_type(ctype + stars, arr)))
        structs.append((m.group(3), fields))
    body = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    body = re.sub(r"(typedef\s+)?enum\s*\w*\s*\{[^}]*\}\s*\w*\s*;", " ", body)
    for m in re.finditer(r"([\w\s\*]+?)\b(blub_\w+)\s*\(([^;{}]*)\)\s*;", body):
        ret, name, args = " ".join(m.group(1).split()), m.group(2), " ".join(m.group(3).split())
        if "typedef" in ret or not ret:
            continue
        params = []
        if args and args != "void":
            for i, a in enumerate(args.split(",")):
                a = a.strip()
                arr = "[]" if re.search(r"\[[^\]]*\]$", a) else ""
                a = re.sub(r"\[[^\]]*\]$", "", a).strip()
                mm = re.match(r"(.+?)(\w+)$", a)
                if mm and mm.group(1).strip() and mm.group(2) not in PRIM and mm.group(2) not in ("blub_fluid", "blub_slab_group", "blub_controller"):
                    ctype, pname = mm.group(1).strip(), mm.group(2)
                else:
                    ctype, pname = a, "arg%d" % i
                if arr:
                    ctype += "*"
                params.append("%s: %s" % (pname, rust_type(ctype)))
        rt = rust_type(ret)
        funcs.append((name, params, "" if rt == "c_void" else " -> " + rt))
    return consts, opaque, structs, funcs


def generate():
    consts, opaque, structs, funcs = parse(open(HEADER).read())
    out = ["```rust", "// GENERATED from include/blubhip.h by tools/gen_rust_ffi.py -- do not edit by hand.", "#![allow(non_camel_case_types, non_upper_case_globals)]",
           "use std::os::raw::{c_char, c_int, c_void};", ""]
    for name, val in consts:
        out.append("pub const %s: c_int = %d;" % (name, val))
    out.append("")
    for name in opaque:
        out.append("#[repr(C)] pub struct %s { _private: [u8; 0] }" % name)
    for name, fields in structs:
        out.append("#[repr(C)] #[derive(Copy, Clone)]")
        out.append("pub struct %s {" % name)
        for f, t in fields:
            out.append("    pub %s: %s," % (f, t))
        out.append("}")
    out += ["", "#[link(name = \"blubhip\")]", "extern \"C\" {"]
    for name, params, ret in funcs:
        out.append("    pub fn %s(%s)%s;" % (name, ", ".join(params), ret))
    out += ["}", "```"]
    return "\n".join(out), [f[0] for f in funcs]


def main():
    block, names = generate()
    if "--write" in sys.argv:
        doc = open(DOC).read()
        a, b = doc.index(BEGIN) + len(BEGIN), doc.index(END)
        open(DOC, "w").write(doc[:a] + "\n" + block + "\n" + doc[b:])
        print("INTEGRATION.md: %d functions" % len(names))
    else:
        print(block)


if __name__ == "__main__":
    main()
