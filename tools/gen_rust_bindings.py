#!/usr/bin/env python3
"""Development tool: regenerate the `extern "C"` block, the status constants and the #[repr(C)] structs of
bindings/stark_mi.rs from include/stark_mi.h (between the BEGIN/END GENERATED markers; the hand-written safe
wrappers below the markers are left alone).  There is no rustc in the image, so the file is checked
structurally by tests/test_rust_binding.py, which parses both files with its own code.

    python3 tools/gen_rust_bindings.py            # rewrites bindings/stark_mi.rs in place
"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "stark_mi.h")
OUT = os.path.join(ROOT, "bindings", "stark_mi.rs")

SCALAR = {"int": "c_int", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64",
          "char": "c_char", "void": "c_void"}
OPAQUE = ("smi_ctx", "smi_tree", "smi_fri_run", "smi_mgpu")


def strip_comments(text):
    return re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def rust_type(ctype):
    """C parameter / return type (already stripped of the name) -> Rust"""
    t = " ".join(ctype.replace("*", " * ").split())
    m = re.fullmatch(r"(const )?(\w+)((?: \*(?: const)?)*)", t)
    if not m:
        raise ValueError(f"cannot translate type {ctype!r}")
    const, base, stars = bool(m.group(1)), m.group(2), m.group(3).count("*")
    rb = SCALAR.get(base, base)
    if stars == 0:
        return rb
    # C reads right to left: `const T *` = pointer to const T; `T **` = pointer to (mutable) pointer to T
    inner = rb
    for level in range(stars):
        is_innermost = level == 0
        inner = ("*const " if (const and is_innermost) else "*mut ") + inner
    return inner


def split_params(params):
    out, depth, cur = [], 0, ""
    for ch in params:
        if ch == "(":
            depth += 1
        if ch == ")":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def parse_param(p):
    """'const uint8_t id[128]' -> (name, rust type)"""
    p = " ".join(p.split())
    m = re.fullmatch(r"(.*?)(\w+)\[(\w+)\]", p)          # array parameter = pointer
    if m:
        return m.group(2), rust_type(m.group(1).strip() + " *")
    m = re.fullmatch(r"(.*?)(\w+)", p)
    return m.group(2), rust_type(m.group(1).strip())


def functions(text):
    out = []
    for m in re.finditer(r"^(?!typedef)([A-Za-z_][\w \*]*?)\b(smi_\w+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.M):
        ret, name, params = m.group(1).strip(), m.group(2), m.group(3).strip()
        ps = [] if params in ("", "void") else [parse_param(p) for p in split_params(params)]
        out.append((name, ps, None if ret == "void" else rust_type(ret)))
    return out


def status_codes(text):
    body = re.search(r"enum\s*\{(.*?)\};", text, flags=re.S).group(1)
    return [(m.group(1), int(m.group(2))) for m in re.finditer(r"(SMI_\w+)\s*=\s*(-?\d+)", body)]


def structs(text):
    """typedef struct { scalar fields } name;  (function-pointer structs are written by hand)"""
    out = []
    for m in re.finditer(r"typedef struct\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        body, name = m.group(1), m.group(2)
        if "(*" in body:
            continue
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            am = re.fullmatch(r"(\w+) (\w+)\[(\d+)\]", decl)
            if am:
                fields.append((am.group(2), f"[{SCALAR[am.group(1)]}; {am.group(3)}]"))
                continue
            ty, names = decl.split(" ", 1)
            for nm in names.split(","):
                fields.append((nm.strip(), SCALAR[ty]))
        out.append((name, fields))
    return out


def generate():
    text = strip_comments(open(HEADER).read())
    lines = ["// BEGIN GENERATED (tools/gen_rust_bindings.py from include/stark_mi.h) -- do not edit by hand"]
    lines.append("/// Status codes (include/stark_mi.h): 0 = ok; -1..-18 mirror a reference panic; the rest are contract or runtime errors.")
    for name, val in status_codes(text):
        lines.append(f"pub const {name}: c_int = {val};")
    id_bytes = re.search(r"#define SMI_MGPU_ID_BYTES (\d+)", text).group(1)
    lines.append(f"pub const SMI_MGPU_ID_BYTES: usize = {id_bytes};")
    lines.append("")
    for o in OPAQUE:
        lines.append(f"#[repr(C)] pub struct {o} {{ _p: [u8; 0] }}")
    lines.append("")
    for name, fields in structs(text):
        lines.append("#[repr(C)] #[derive(Clone, Copy, Debug)]")
        lines.append(f"pub struct {name} {{")
        for fn_, ty in fields:
            lines.append(f"    pub {fn_}: {ty},")
        lines.append("}")
    lines.append("")
    lines.append("/// smi_mgpu_coll: three caller-supplied collectives (device pointers; 0 = ok), how another transport is plugged in")
    lines.append("#[repr(C)] #[derive(Clone, Copy)]")
    lines.append("pub struct smi_mgpu_coll {")
    lines.append("    pub user: *mut c_void,")
    lines.append("    pub all_gather: Option<unsafe extern \"C\" fn(user: *mut c_void, d_send: *const c_void, d_recv: *mut c_void, bytes_per_rank: usize) -> c_int>,")
    lines.append("    pub exchange: Option<unsafe extern \"C\" fn(user: *mut c_void, n_send: c_int, send_peer: *const c_int, d_send: *const *mut c_void, send_bytes: *const usize,")
    lines.append("                                                n_recv: c_int, recv_peer: *const c_int, d_recv: *const *mut c_void, recv_bytes: *const usize) -> c_int>,")
    lines.append("    pub all_reduce_sum_u8: Option<unsafe extern \"C\" fn(user: *mut c_void, d_buf: *mut c_void, bytes: usize) -> c_int>,")
    lines.append("}")
    lines.append("")
    lines.append('#[link(name = "starkmi")]')
    lines.append('extern "C" {')
    for name, ps, ret in functions(text):
        args = ", ".join(f"{n}: {t}" for n, t in ps)
        lines.append(f"    pub fn {name}({args})" + (f" -> {ret};" if ret else ";"))
    lines.append("}")
    lines.append("// END GENERATED")
    return "\n".join(lines)


def main():
    gen = generate()
    src = open(OUT).read()
    a, b = src.index("// BEGIN GENERATED"), src.index("// END GENERATED") + len("// END GENERATED")
    open(OUT, "w").write(src[:a] + gen + src[b:])
    print(f"{OUT}: {gen.count('pub fn ')} declarations")


if __name__ == "__main__":
    main()
