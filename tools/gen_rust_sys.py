#!/usr/bin/env python3
"""Writes rust/oar-mi355x-sys/src/lib.rs from include/oar_mi355x.h (a purpose-built bindgen: this image has no Rust
toolchain and no libclang bindings, and the header is plain C89 declarations).

    python tools/gen_rust_sys.py            # rewrite the crate's lib.rs
    python tools/gen_rust_sys.py --check    # exit 1 when the committed lib.rs differs from what the header yields

`parse_header()` is also what tests/test_rust_bindings_cpu.py uses to diff the crate against the header and against
the symbols the built libOarMi355x.so exports.
"""
from __future__ import annotations

import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "oar_mi355x.h")
OUT = os.path.join(ROOT, "rust", "oar-mi355x-sys", "src", "lib.rs")

PRIM = {"uint8_t": "u8", "int8_t": "i8", "uint16_t": "u16", "int16_t": "i16", "int32_t": "i32", "uint32_t": "u32", "int64_t": "i64",
        "uint64_t": "u64", "float": "f32", "double": "f64", "size_t": "usize", "int": "c_int", "char": "c_char", "void": "c_void"}


def strip_comments(text: str) -> str:
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)                 # preprocessor lines
    text = text.replace('extern "C" {', " ")
    return text


def statements(text: str):
    """Top-level `;`-terminated statements (braces of struct / enum bodies are kept inside their statement)."""
    depth, cur, out = 0, [], []
    for ch in text:
        if ch == "{":
            depth += 1
        elif ch == "}":
            depth -= 1
            if depth < 0:            # the closing brace of extern "C"
                depth = 0
                continue
        if ch == ";" and depth == 0:
            s = " ".join("".join(cur).split())
            if s:
                out.append(s)
            cur = []
        else:
            cur.append(ch)
    return out


def rust_type(ctype: str) -> str:
    """'const uint8_t* const*' -> '*const *const u8'"""
    t = ctype.replace("*", " * ").split()
    # base type = everything up to the first '*'
    base, i, base_const = None, 0, False
    while i < len(t) and t[i] != "*":
        if t[i] == "const":
            base_const = True
        elif t[i] in ("struct", "unsigned", "signed"):
            raise ValueError("unsupported C type: " + ctype)
        else:
            base = t[i]
        i += 1
    out = PRIM.get(base, base)
    const_next = base_const
    while i < len(t):
        assert t[i] == "*", ctype
        i += 1
        ptr_const = False
        if i < len(t) and t[i] == "const":
            ptr_const = True
            i += 1
        out = ("*const " if const_next else "*mut ") + out
        const_next = ptr_const
    return out


def parse_decl(decl: str):
    """'const float box[8]' -> ('box', '*const f32' as parameter / '[f32; 8]' as field, 8)"""
    m = re.match(r"^(.*?)([A-Za-z_][A-Za-z0-9_]*)\s*(\[(\d+)\])?$", decl.strip())
    if not m:
        raise ValueError("cannot parse declaration: " + decl)
    ctype, name, _, n = m.group(1).strip(), m.group(2), m.group(3), m.group(4)
    return ctype, name, int(n) if n else None


def parse_header(path: str = HEADER):
    raw = open(path).read()
    defines = [(m.group(1), int(m.group(2), 0)) for m in re.finditer(r"^#define (OAR_\w+) (0x[0-9A-Fa-f]+|\d+)u?\s*$", raw, flags=re.M)]
    text = strip_comments(raw)
    enums, opaque, structs, fnptrs, funcs = [], [], [], [], []
    for st in statements(text):
        m = re.match(r"^typedef enum \{(.*)\} (\w+)$", st)
        if m:
            items = []
            for it in m.group(1).split(","):
                it = it.strip()
                if not it:
                    continue
                k, v = [s.strip() for s in it.split("=")]
                items.append((k, int(v)))
            enums.append((m.group(2), items))
            continue
        m = re.match(r"^typedef struct (\w+) (\w+)$", st)
        if m:
            assert m.group(1) == m.group(2)
            opaque.append(m.group(2))
            continue
        m = re.match(r"^typedef struct \{(.*)\} (\w+)$", st)
        if m:
            fields = []
            for f in m.group(1).split(";"):
                f = f.strip()
                if not f:
                    continue
                first, *rest = [s.strip() for s in f.split(",")]
                ctype, name, n = parse_decl(first)
                fields.append((name, ctype, n))
                for r in rest:                                   # `uint32_t a, b;`: the others share the base type
                    _, name2, n2 = parse_decl(ctype + " " + r)
                    fields.append((name2, ctype, n2))
            structs.append((m.group(2), fields))
            continue
        m = re.match(r"^typedef (\w[\w\s\*]*?)\s*\(\*(\w+)\)\s*\((.*)\)$", st)
        if m:
            fnptrs.append((m.group(2), m.group(1).strip(), [parse_decl(p) for p in m.group(3).split(",")]))
            continue
        m = re.match(r"^([\w\s\*]+?)\s*\b(\w+)\s*\((.*)\)$", st)
        if m:
            params = [] if m.group(3).strip() == "void" else [parse_decl(p) for p in m.group(3).split(",")]
            funcs.append((m.group(2), m.group(1).strip(), params))
            continue
        raise ValueError("unrecognised declaration in header: " + st[:120])
    return {"enums": enums, "opaque": opaque, "structs": structs, "fnptrs": fnptrs, "funcs": funcs, "defines": defines}


RUST_KEYWORDS = {"as", "box", "break", "const", "continue", "crate", "dyn", "else", "enum", "extern", "false", "fn", "for", "if", "impl", "in",
                 "let", "loop", "match", "mod", "move", "mut", "pub", "ref", "return", "self", "static", "struct", "super", "trait", "true",
                 "type", "unsafe", "use", "where", "while", "async", "await", "abstract", "become", "do", "final", "macro", "override", "priv",
                 "try", "typeof", "unsized", "virtual", "yield"}


def ident(name: str) -> str:
    """parameter names are not part of the ABI: a C name that is a Rust keyword gets a trailing underscore"""
    return name + "_" if name in RUST_KEYWORDS else name


def param_type(ctype: str, n):
    if n is None:
        return rust_type(ctype)
    return rust_type(ctype + "*")          # `const float box[8]` decays to `const float*`


def ret_type(ctype: str) -> str:
    return "" if ctype == "void" else " -> " + rust_type(ctype)


def generate(h) -> str:
    L = []
    w = L.append
    w("//! Raw FFI bindings to `libOarMi355x.so` (the MI355X / gfx950 drop-in for the det+rec hot path of oar-ocr).")
    w("//!")
    w("//! GENERATED from `include/oar_mi355x.h` by `tools/gen_rust_sys.py` -- do not edit by hand; the header carries the")
    w("//! documentation of every item, including the reference interface (file:line) each entry point replaces.")
    w("//! `tests/test_rust_bindings_cpu.py` fails when this file and the header (or the symbols the built library exports)")
    w("//! drift apart.")
    w("#![allow(non_camel_case_types, non_snake_case, clippy::too_many_arguments)]")
    w("#![no_std]")
    w("")
    w("use core::ffi::{c_char, c_int, c_void};")
    w("")
    for name, items in h["enums"]:
        w(f"pub type {name} = c_int;")
        for k, v in items:
            w(f"pub const {k}: {name} = {v};")
        w("")
    for name, value in h["defines"]:
        w(f"pub const {name}: u32 = 0x{value:08X};")
    if h["defines"]:
        w("")
    for name in h["opaque"]:
        w("#[repr(C)]")
        w(f"pub struct {name} {{")
        w("    _private: [u8; 0],")
        w("    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,")
        w("}")
        w("")
    for name, ret, params in h["fnptrs"]:
        ps = ", ".join(f"{ident(pn)}: {param_type(ct, n)}" for ct, pn, n in params)
        w(f"pub type {name} = Option<unsafe extern \"C\" fn({ps}){ret_type(ret)}>;")
        w("")
    for name, fields in h["structs"]:
        w("#[repr(C)]")
        w("#[derive(Debug, Clone, Copy)]")
        w(f"pub struct {name} {{")
        for fn_, ct, n in fields:
            assert fn_ not in RUST_KEYWORDS, f"{name}.{fn_}: rename the field in the header (Rust keyword)"
            rt = rust_type(ct)
            w(f"    pub {fn_}: {('[' + rt + '; ' + str(n) + ']') if n is not None else rt},")
        w("}")
        w("")
    w('#[link(name = "OarMi355x")]')
    w('unsafe extern "C" {')
    for name, ret, params in h["funcs"]:
        ps = ", ".join(f"{ident(pn)}: {param_type(ct, n)}" for ct, pn, n in params)
        notes = [f"{ident(pn)}: [{PRIM.get(ct.replace('const', '').strip(), ct)}; {n}]" for ct, pn, n in params if n is not None]
        if notes:
            w("    /// fixed-length arrays: " + ", ".join(notes))
        w(f"    pub fn {name}({ps}){ret_type(ret)};")
    w("}")
    w("")
    return "\n".join(L)


def main():
    text = generate(parse_header())
    if "--check" in sys.argv:
        have = open(OUT).read() if os.path.exists(OUT) else ""
        if have != text:
            print("rust/oar-mi355x-sys/src/lib.rs is stale: run python tools/gen_rust_sys.py", file=sys.stderr)
            sys.exit(1)
        return
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        f.write(text)
    print(OUT, len(text.splitlines()), "lines")


if __name__ == "__main__":
    main()
