#!/usr/bin/env python3
"""Generate the Rust `extern "C"` binding of include/wrk_hip.h (the block INTEGRATION.md section 1 shows).

The binding a maintainer adds on the reference side (src/backend/hip.rs) must match the header exactly -- a `*const WrkBuf` where the
header takes a `const wrk_tensor*` is undefined behaviour that no compiler reports (INTEGRATION.md had drifted that way).
So the block is GENERATED from the header, INTEGRATION.md carries the output between two markers, and tests/test_abi_host.py regenerates it
and compares.  `python tools/gen_rust_binding.py --write` refreshes INTEGRATION.md after a header change.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "wrk_hip.h")
DOC = os.path.join(ROOT, "INTEGRATION.md")
BEGIN, END = "<!-- BEGIN GENERATED: tools/gen_rust_binding.py -->", "<!-- END GENERATED -->"

SCALARS = {"int32_t": "i32", "uint32_t": "u32", "uint16_t": "u16", "uint8_t": "u8", "size_t": "usize", "float": "f32", "int": "i32",
           "void": "c_void", "char": "c_char"}


def strip_comments(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def rust_name(c):
    """wrk_v7_model_desc -> WrkV7ModelDesc"""
    return "".join(p.capitalize() for p in c.split("_"))


def rust_type(base, const, depth):
    t = SCALARS.get(base) or rust_name(base)
    for i in range(depth):
        # innermost pointer carries the C const; outer levels are out-parameters
        t = ("*const " if (const and i == 0) else "*mut ") + t
    return t


def parse_decl(decl):
    """'const wrk_buf* w' / 'wrk_ctx** out' / 'uint32_t k' -> (name, base, const, depth)"""
    decl = " ".join(decl.split())
    const = bool(re.search(r"\bconst\b", decl))
    decl = re.sub(r"\bconst\b", "", decl).strip()
    m = re.match(r"^(\w+)\s*(\**)\s*(\w+)?$", decl.replace(" *", "*").replace("* ", "*"))
    if not m:
        raise ValueError(f"cannot parse declaration: {decl!r}")
    return m.group(3), m.group(1), const, len(m.group(2))


def parse_header(text):
    text = strip_comments(text)
    opaque, structs, funcs = [], [], []
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s+(\w+)\s*;", text):
        opaque.append(m.group(2))
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for stmt in m.group(2).split(";"):
            stmt = " ".join(stmt.split())
            if not stmt:
                continue
            const = bool(re.search(r"\bconst\b", stmt))
            stmt = re.sub(r"\bconst\b", "", stmt).strip()
            base, rest = stmt.split(None, 1) if " " in stmt and not stmt.split()[0].endswith("*") else (stmt.split("*")[0].strip(), stmt[len(stmt.split("*")[0]):])
            base = base.rstrip("*")
            if stmt.startswith(base + "*"):
                rest = stmt[len(base):]
            for d in rest.split(","):
                d = d.strip()
                depth = d.count("*")
                d = d.replace("*", "").strip()
                am = re.match(r"^(\w+)\[(\d+)\]$", d)
                name, arr = (am.group(1), int(am.group(2))) if am else (d, 0)
                fields.append((name, base, const, depth, arr))
        structs.append((m.group(3), fields))
    for m in re.finditer(r"^\s*([\w\s\*]+?)\s*\b(wrk_\w+)\s*\(([^;{]*?)\)\s*;", text, flags=re.M | re.S):
        ret = " ".join(m.group(1).split())
        args = " ".join(m.group(3).split())
        params = [] if args in ("", "void") else [parse_decl(a) for a in args.split(",")]
        rconst = "const" in ret
        rbase = re.sub(r"\bconst\b", "", ret).replace("*", "").strip()
        funcs.append((m.group(2), (rbase, rconst, ret.count("*")), params))
    return opaque, structs, funcs


def generate(text):
    opaque, structs, funcs = parse_header(text)
    out = ["// src/backend/hip.rs -- generated from include/wrk_hip.h by tools/gen_rust_binding.py; do not edit by hand",
           "use std::os::raw::{c_char, c_void};", ""]
    out.append(" ".join(f"pub enum {rust_name(o)} {{}}" for o in opaque) + "      // opaque handles")
    for name, fields in structs:
        out.append("#[repr(C)]")
        out.append(f"pub struct {rust_name(name)} {{")
        for fname, base, const, depth, arr in fields:
            t = rust_type(base, const, depth)
            if arr:
                t = f"[{t}; {arr}]"
            out.append(f"    pub {fname}: {t},")
        out.append("}")
    out += ["", '#[link(name = "wrk_hip")]', 'extern "C" {']
    for name, (rbase, rconst, rdepth), params in funcs:
        ps = ", ".join(f"{'r#' if p[0] in ('type', 'ref', 'in', 'box') else ''}{p[0]}: {rust_type(p[1], p[2], p[3])}" for p in params)
        ret = rust_type(rbase, rconst, rdepth)
        tail = "" if (rbase == "void" and rdepth == 0) else f" -> {ret}"
        out.append(f"    pub fn {name}({ps}){tail};")
    out.append("}")
    return "\n".join(out) + "\n"


def doc_block(doc):
    a, b = doc.index(BEGIN), doc.index(END)
    body = doc[a + len(BEGIN):b]
    m = re.search(r"```rust\n(.*?)```", body, flags=re.S)
    return m.group(1) if m else None


def main():
    text = open(HEADER).read()
    gen = generate(text)
    if "--write" in sys.argv:
        doc = open(DOC).read()
        a, b = doc.index(BEGIN), doc.index(END)
        doc = doc[:a + len(BEGIN)] + "\n```rust\n" + gen + "```\n" + doc[b:]
        open(DOC, "w").write(doc)
        print(f"INTEGRATION.md refreshed ({gen.count(chr(10))} lines)")
    else:
        sys.stdout.write(gen)


if __name__ == "__main__":
    main()
