"""CPU: the shipped Rust side of the boundary (bindings/stark_mi.rs) against the C header it binds
(include/stark_mi.h).  There is no rustc in the image, so the bar is structural: every function the header
declares is declared in the `extern "C"` block with the same arity, and every parameter / return value has the
same scalar width, pointer depth and pointee constness; the #[repr(C)] structs have the header's fields in the
header's order (sizes cross-checked with ctypes); the status constants are equal; the safe wrappers only call
declared entry points.  Both files are parsed here with code of this test's own (not tools/gen_rust_bindings.py);
a last test makes sure the generated block is not stale."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "stark_mi.h")
RUST = os.path.join(ROOT, "bindings", "stark_mi.rs")

# canonical scalar classes: (kind, bits) -- `int` is c_int on both sides, size_t is usize
C_SCALAR = {"int": "c_int", "uint8_t": "u8", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "double": "f64",
            "char": "c_char", "void": "void"}
R_SCALAR = {"c_int": "c_int", "u8": "u8", "u32": "u32", "u64": "u64", "usize": "usize", "f64": "f64", "c_char": "c_char",
            "c_void": "void"}
OPAQUE = {"smi_ctx", "smi_tree", "smi_fri_run", "smi_mgpu", "smi_fri_cfg", "smi_stark_cfg", "smi_kernel_time", "smi_mgpu_coll"}


def _c_text():
    return re.sub(r"/\*.*?\*/", " ", open(HEADER).read(), flags=re.S)


def _c_type(tokens):
    """['const', 'uint8_t', '*', '*'] -> (base class, pointer depth, innermost pointee const)"""
    toks = [t for t in tokens if t]
    const_first = bool(toks) and toks[0] == "const"
    if const_first:
        toks = toks[1:]
    base = toks[0]
    depth = sum(1 for t in toks[1:] if t == "*")
    cls = C_SCALAR.get(base, base if base in OPAQUE else None)
    assert cls is not None, f"unknown C type {tokens}"
    return cls, depth, (const_first if depth else False)


def c_functions():
    text = _c_text()
    out = {}
    # a declaration: everything from a line start up to ';' that contains 'smi_xxx(' and is not a typedef / struct body
    for decl in re.findall(r"(?m)^(?:const\s+)?[a-z_0-9]+[\s\*]+smi_[a-z0-9_]+\s*\([^;]*\)\s*;", text):
        head, params = decl[:decl.index("(")], decl[decl.index("(") + 1:decl.rindex(")")]
        name = re.search(r"(smi_[a-z0-9_]+)\s*$", head).group(1)
        ret_tokens = re.findall(r"\w+|\*", head[:head.rindex(name)])
        ret = None if ret_tokens == ["void"] else _c_type(ret_tokens)
        ps = []
        if params.strip() not in ("", "void"):
            for p in params.split(","):
                toks = re.findall(r"\w+|\*|\[|\]", p)
                if "[" in toks:                                       # `uint8_t root[32]` is a pointer parameter
                    toks = toks[:toks.index("[")] + ["*"]
                    pname = toks[-2]
                    toks = toks[:-2] + ["*"]
                else:
                    pname = toks[-1]
                    toks = toks[:-1]
                ps.append((pname, _c_type(toks)))
        out[name] = (ps, ret)
    return out


def _r_type(s):
    s = s.strip()
    depth, inner_const = 0, False
    while True:
        m = re.match(r"\*(const|mut)\s+", s)
        if not m:
            break
        depth += 1
        inner_const = m.group(1) == "const"      # the last pointer prefix is the innermost pointer
        s = s[m.end():]
    cls = R_SCALAR.get(s, s if s in OPAQUE else None)
    assert cls is not None, f"unknown Rust type {s!r}"
    return cls, depth, (inner_const if depth else False)


def rust_extern_block():
    src = open(RUST).read()
    m = re.search(r'extern "C" \{(.*?)\n\}', src, flags=re.S)
    assert m, "no extern block"
    return m.group(1)


def rust_functions():
    out = {}
    for m in re.finditer(r"pub fn (smi_\w+)\((.*?)\)(?:\s*->\s*([^;]+))?;", rust_extern_block(), flags=re.S):
        name, params, ret = m.group(1), m.group(2), m.group(3)
        ps = []
        for p in [x for x in params.split(",") if x.strip()]:
            pname, ty = p.split(":", 1)
            ps.append((pname.strip(), _r_type(ty)))
        out[name] = (ps, _r_type(ret) if ret else None)
    return out


def test_every_entry_point_is_declared_with_the_headers_signature():
    c, r = c_functions(), rust_functions()
    import stark_rs_amd
    assert sorted(c) == stark_rs_amd.declared_symbols()              # this test's C parser sees what the ABI test sees
    assert len(c) >= 75
    missing = sorted(set(c) - set(r))
    extra = sorted(set(r) - set(c))
    assert not missing, f"declared in stark_mi.h but not in stark_mi.rs: {missing}"
    assert not extra, f"declared in stark_mi.rs but not in stark_mi.h: {extra}"
    for name in sorted(c):
        (cp, cr), (rp, rr) = c[name], r[name]
        assert len(cp) == len(rp), f"{name}: arity {len(cp)} in C, {len(rp)} in Rust"
        assert cr == rr, f"{name}: return {cr} in C, {rr} in Rust"
        for (cn, ct), (rn, rt) in zip(cp, rp):
            assert ct == rt, f"{name}({cn}): {ct} in C, {rt} in Rust ({rn})"
            assert cn == rn or cn.lower() == rn.lower(), f"{name}: parameter {cn} is called {rn} in Rust"


def _c_structs():
    out = {}
    for m in re.finditer(r"typedef struct\s*\{(.*?)\}\s*(\w+)\s*;", _c_text(), flags=re.S):
        body, name = m.group(1), m.group(2)
        if "(*" in body:
            continue
        fields = []
        for decl in [" ".join(d.split()) for d in body.split(";") if d.strip()]:
            am = re.fullmatch(r"(\w+) (\w+)\[(\d+)\]", decl)
            if am:
                fields.append((am.group(2), C_SCALAR[am.group(1)], int(am.group(3))))
                continue
            ty, names = decl.split(" ", 1)
            fields += [(nm.strip(), C_SCALAR[ty], 1) for nm in names.split(",")]
        out[name] = fields
    return out


def _r_structs():
    out = {}
    src = open(RUST).read()
    for m in re.finditer(r"#\[repr\(C\)\][^\n]*\npub struct (\w+) \{\n(.*?)\n\}", src, flags=re.S):
        fields = []
        for line in m.group(2).splitlines():
            fm = re.match(r"\s*pub (\w+): (.+?),\s*$", line)
            if not fm:
                continue
            am = re.fullmatch(r"\[(\w+); (\d+)\]", fm.group(2))
            if am:
                fields.append((fm.group(1), R_SCALAR[am.group(1)], int(am.group(2))))
            elif fm.group(2) in R_SCALAR:
                fields.append((fm.group(1), R_SCALAR[fm.group(2)], 1))
            else:
                fields.append((fm.group(1), fm.group(2), 1))
        out[m.group(1)] = fields
    return out


def test_repr_c_structs_have_the_headers_layout():
    c, r = _c_structs(), _r_structs()
    assert set(c) == {"smi_kernel_time", "smi_fri_cfg", "smi_stark_cfg"}
    for name, fields in c.items():
        assert r.get(name) == fields, f"{name}: {fields} in C, {r.get(name)} in Rust"
    # sizes as the C compiler lays them out (ctypes mirrors of the product) against the Rust field lists
    from stark_rs_amd import _lib
    width = {"c_int": 4, "u8": 1, "u32": 4, "u64": 8, "usize": 8, "f64": 8, "c_char": 1}
    def size_of(fields):
        off, align = 0, 1
        for _n, ty, cnt in fields:
            w = width[ty]
            off = (off + w - 1) // w * w + w * cnt
            align = max(align, w)
        return (off + align - 1) // align * align
    assert size_of(r["smi_fri_cfg"]) == C.sizeof(_lib.FriCfg) == 40
    assert size_of(r["smi_stark_cfg"]) == C.sizeof(_lib.StarkCfg) == 48
    assert size_of(r["smi_kernel_time"]) == C.sizeof(_lib.KernelTime) == 88
    # the collective table: user pointer + three function pointers, in the header's order
    src = open(RUST).read()
    coll = re.search(r"pub struct smi_mgpu_coll \{(.*?)\n\}", src, flags=re.S).group(1)
    assert re.findall(r"pub (\w+):", coll) == ["user", "all_gather", "exchange", "all_reduce_sum_u8"]
    hdr = re.search(r"typedef struct\s*\{([^}]*?\(\*[^}]*?)\}\s*smi_mgpu_coll", _c_text(), flags=re.S).group(1)
    assert re.findall(r"\(\*(\w+)\)", hdr) == ["all_gather", "exchange", "all_reduce_sum_u8"]
    for fn_name in ("all_gather", "exchange", "all_reduce_sum_u8"):
        c_params = re.search(r"\(\*" + fn_name + r"\)\s*\((.*?)\)\s*;", hdr, flags=re.S).group(1).split(",")
        r_params = re.search(r"pub " + fn_name + r": Option<unsafe extern \"C\" fn\((.*?)\) -> c_int>", coll, flags=re.S).group(1).split(",")
        assert len(c_params) == len(r_params), fn_name


def test_status_constants_are_the_headers():
    enum = re.search(r"enum\s*\{(.*?)\};", _c_text(), flags=re.S).group(1)
    c = {m.group(1): int(m.group(2)) for m in re.finditer(r"(SMI_\w+)\s*=\s*(-?\d+)", enum)}
    r = {m.group(1): int(m.group(2)) for m in re.finditer(r"pub const (SMI_(?:OK|ERR_\w+)): c_int = (-?\d+);", open(RUST).read())}
    assert c == r and len(c) >= 28
    assert re.search(r"pub const SMI_MGPU_ID_BYTES: usize = 128;", open(RUST).read())


def test_safe_wrappers_call_declared_entry_points_only():
    src = open(RUST).read()
    hand = src[src.index("// END GENERATED"):]
    used = set(re.findall(r"\b(smi_[a-z0-9_]+)\s*\(", hand))
    declared = set(rust_functions())
    assert used <= declared, sorted(used - declared)
    # the methods INTEGRATION.md's table puts behind the boundary are all wrapped
    for fn_name in ("smi_intt", "smi_coset_ntt", "smi_poly_scale", "smi_poly_mul", "smi_poly_div", "smi_hash_leaves",
                    "smi_hash_combine_pairs", "smi_merkle_new", "smi_merkle_commit", "smi_merkle_open", "smi_merkle_free",
                    "smi_fri_fold", "smi_fri_prove", "smi_fri_verify", "smi_lde", "smi_trace_pack", "smi_domain_is_geometric",
                    "smi_status_string", "smi_ctx_create", "smi_ctx_destroy"):
        assert fn_name in used, fn_name
    # every unsafe call's status goes through check(): no wrapper drops a status on the floor
    for m in re.finditer(r"unsafe \{\s*(smi_\w+)\(", hand):
        if m.group(1) in ("smi_ctx_destroy", "smi_merkle_free", "smi_free", "smi_merkle_num_leaves", "smi_status_string", "smi_last_error"):
            continue
        before = hand[max(0, m.start() - 60):m.start()]
        assert "check(" in before or "match " in before or "let st = " in before, (m.group(1), before)


def test_generated_block_is_current():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_bindings", os.path.join(ROOT, "tools", "gen_rust_bindings.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    src = open(RUST).read()
    a, b = src.index("// BEGIN GENERATED"), src.index("// END GENERATED") + len("// END GENERATED")
    assert src[a:b] == gen.generate(), "bindings/stark_mi.rs is stale: run python3 tools/gen_rust_bindings.py"
