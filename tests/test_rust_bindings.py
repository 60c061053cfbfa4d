"""rust/ecgpu-shim/src/sys.rs against include/ecgpu.h, mechanically.

There is no Rust toolchain in the image, so the `extern "C"` block a Rust host links through is never compiled
here.  What CAN be checked without rustc: every function sys.rs declares exists in the header with the same number of
parameters, every parameter and the return value have the C type the Rust type maps to (width, signedness, pointer depth and
constness), every constant has the header's value, and the safe wrappers in lib.rs call only functions sys.rs declares."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return re.sub(r"//[^\n]*", "", src)


def header_prototypes():
    src = _strip_c(open(os.path.join(ROOT, "include", "ecgpu.h")).read())
    out = {}
    for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(ecgpu_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        args = " ".join(args.split())
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        out[name] = (" ".join(ret.split()), params)
    return out


def header_constants():
    vals = {}
    for f in ("ecgpu.h", "ecgpu_status.h"):
        src = _strip_c(open(os.path.join(ROOT, "include", f)).read())
        for name, v in re.findall(r"#define[ \t]+(ECGPU_[A-Z0-9_]+)[ \t]+([^\n]+)", src):
            v = v.strip().strip("()").rstrip("uUlL")
            try:
                vals[name] = int(v, 0)
            except ValueError:
                pass
    return vals


# the C type of a parameter, reduced to (base, pointer depth, const pointee)
_C_BASE = {"int": "i32", "unsigned": "u32", "unsigned int": "u32", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64",
           "int64_t": "i64", "size_t": "usize", "uint8_t": "u8", "char": "c_char", "double": "f64", "void": "void",
           "long long": "i64"}


def c_type(decl, is_param=True):
    decl = decl.strip()
    arr = re.search(r"\[\d*\]\s*$", decl)  # `double ms[4]` is a pointer
    decl = re.sub(r"\[\d*\]\s*$", "", decl)
    depth = decl.count("*") + (1 if arr else 0)
    decl = decl.replace("*", " ")
    toks = decl.split()
    const = "const" in toks
    toks = [t for t in toks if t not in ("const", "struct")]
    if is_param and len(toks) > 1 and " ".join(toks[:-1]) in _C_BASE or is_param and len(toks) > 1 and toks[-2].endswith("_t") \
            or is_param and len(toks) > 1 and toks[0] in ("ecgpu_ssz_type",):
        toks = toks[:-1]  # the parameter's name
    base = " ".join(toks)
    if base == "ecgpu_stream_t":
        return ("void", depth + 1, False)
    return (_C_BASE.get(base, base), depth, const and depth > 0)


_RS_BASE = {"c_int": "i32", "c_char": "c_char", "c_void": "void", "core::ffi::c_void": "void", "u8": "u8", "u32": "u32", "i32": "i32",
            "u64": "u64", "i64": "i64", "usize": "usize", "f64": "f64"}


def rs_type(t):
    t = t.strip()
    depth, const = 0, False
    first = True
    while t.startswith("*"):
        m = re.match(r"\*(const|mut)\s+", t)
        assert m, t
        if first:
            pass
        const = m.group(1) == "const"  # constness of the INNERMOST pointee is what the C `const T*` states
        depth += 1
        t = t[m.end():]
        first = False
    if t == "ecgpu_stream_t":
        return ("void", depth + 1, False)
    return (_RS_BASE.get(t, t), depth, const)


def rust_externs():
    src = open(os.path.join(ROOT, "rust", "ecgpu-shim", "src", "sys.rs")).read()
    src = re.sub(r"//[^\n]*", "", src)
    block = src[src.index('extern "C" {'):]
    out = {}
    for name, args, ret in re.findall(r"pub fn (ecgpu_[a-z0-9_]+)\s*\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
        params = [a.strip() for a in " ".join(args.split()).split(",") if a.strip()]
        out[name] = ((ret or "").strip(), [p.split(":", 1)[1].strip() for p in params], [p.split(":", 1)[0].strip() for p in params])
    return out, src


def test_every_rust_extern_is_a_header_prototype_with_the_same_types():
    protos = header_prototypes()
    externs, _ = rust_externs()
    assert len(externs) >= 60
    for name, (ret, ptypes, pnames) in externs.items():
        assert name in protos, f"sys.rs declares {name}, include/ecgpu.h does not"
        c_ret, c_params = protos[name]
        assert len(c_params) == len(ptypes), (name, c_params, ptypes)
        want_ret = c_type(c_ret, is_param=False)
        got_ret = rs_type(ret) if ret else ("void", 0, False)
        assert got_ret == want_ret, (name, "return", ret, c_ret)
        for c, r, rn in zip(c_params, ptypes, pnames):
            assert rs_type(r) == c_type(c), (name, rn, r, c)


def test_rust_parameter_names_follow_the_header():
    """Positional mistakes between same-typed neighbours (msgs / sigs, offsets / data_off) do not change a type: the NAMES must
    agree too (the header's `d_` prefix for device pointers included)."""
    protos = header_prototypes()
    externs, _ = rust_externs()
    for name, (_, ptypes, pnames) in externs.items():
        c_names = [re.sub(r"\[\d*\]$", "", c.strip()).replace("*", " ").split()[-1] for c in protos[name][1]]
        assert c_names == pnames, (name, c_names, pnames)


def test_rust_constants_have_the_headers_values():
    vals = header_constants()
    _, src = rust_externs()
    consts = re.findall(r"pub const (ECGPU_[A-Z0-9_]+)\s*:\s*[a-z0-9_]+\s*=\s*(-?(?:0x[0-9a-fA-F]+|\d+))\s*;", src)
    assert len(consts) >= 30
    for name, v in consts:
        assert name in vals, name
        assert int(v, 0) == vals[name], (name, v, vals[name])


def test_the_safe_wrappers_call_only_declared_functions():
    externs, _ = rust_externs()
    lib = open(os.path.join(ROOT, "rust", "ecgpu-shim", "src", "lib.rs")).read()
    lib = re.sub(r"//[^\n]*", "", lib)
    called = set(re.findall(r"\b(?:sys::)?(ecgpu_[a-z0-9_]+)\s*\(", lib))
    assert called, "lib.rs calls nothing?"
    missing = sorted(c for c in called if c not in externs)
    assert not missing, missing
    used_consts = set(re.findall(r"\b(?:sys::)?(ECGPU_[A-Z0-9_]+)\b", lib))
    _, src = rust_externs()
    declared = set(re.findall(r"pub const (ECGPU_[A-Z0-9_]+)", src))
    assert not sorted(used_consts - declared), sorted(used_consts - declared)


def test_the_patches_use_only_names_the_shim_defines():
    """rust/patches/*.patch is what a maintainer applies to the reference: every `ecgpu_shim::…` path and every `m.<hook>(` of the
    mirror it mentions must be a public item of the crate."""
    import glob
    shim = open(os.path.join(ROOT, "rust", "ecgpu-shim", "src", "lib.rs")).read() + open(os.path.join(ROOT, "rust", "ecgpu-shim", "src", "sys.rs")).read()
    public = set(re.findall(r"pub (?:fn|struct|enum|mod|const|type)\s+([A-Za-z_][A-Za-z0-9_]*)", shim))
    variants = set(re.findall(r"^\s+([A-Z][A-Za-z]+)(?:\(|,|\s*$)", shim, flags=re.M))
    used_paths, used_hooks = set(), set()
    for p in glob.glob(os.path.join(ROOT, "rust", "patches", "*.patch")):
        text = "\n".join(l[1:] for l in open(p).read().splitlines() if l.startswith("+") and not l.startswith("+++"))
        used_paths |= set(re.findall(r"ecgpu_shim((?:::[A-Za-z_][A-Za-z0-9_]*)+)", text))
        used_hooks |= set(re.findall(r"\bm\.([a-z_]+)\(", text))
    assert len(used_paths) >= 20 and len(used_hooks) >= 10
    for path in used_paths:
        for part in path.split("::")[1:]:
            assert part in public or part in variants, (path, part)
    for h in used_hooks:
        assert re.search(r"pub fn %s\s*\(\s*&mut self" % h, shim), h
