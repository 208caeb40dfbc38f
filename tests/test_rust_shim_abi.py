"""The Rust shim cannot be compiled in this image (no rustc / cargo: SURVEY.md fact 1), so its `extern "C"` block and its
`#[repr(C)]` structs are checked against include/fractal_hip.h by machine instead: every function the shim declares must
exist in the header with the same arity and the same parameter / return types (c_int <-> int, usize <-> size_t, u32 <->
uint32_t, `*const fr_config` <-> `const fr_config *`, ...), and every struct must list the header's fields in the header's
order with the same types.  Plus: the surface the reference's call sites need (get_recursive_pixel calc/src/lib.rs:199,
recursive :245, get_image src/lib.rs:253, the GUI's RGBA frame src/gui.rs:71-72) has externs AND safe wrappers."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "fractal_hip.h")
SHIM = os.path.join(ROOT, "rust", "fractal-hip-sys", "src", "lib.rs")

C_SCALARS = {"int": "c_int", "uint32_t": "u32", "int32_t": "i32", "uint64_t": "u64", "uint8_t": "u8", "size_t": "usize",
             "double": "f64", "float": "f32", "char": "c_char", "void": "c_void"}


def strip_c_comments(text):
    return re.sub(r"/\*.*?\*/", " ", text, flags=re.S)


def c_type_to_rust(t):
    """`const fr_config *` -> `*const fr_config`; `uint32_t` -> `u32`; `void *` -> `*mut c_void`."""
    t = " ".join(t.replace("*", " * ").split())
    if t.endswith("*"):
        base = t[:-1].strip()
        const = base.startswith("const ")
        base = base[6:] if const else base
        assert "*" not in base, "pointer to pointer: extend the checker (%r)" % t
        return ("*const " if const else "*mut ") + C_SCALARS.get(base, base)
    assert not t.startswith("const "), t
    return C_SCALARS.get(t, t)


def parse_c_params(args):
    args = args.strip()
    if args in ("", "void"):
        return []
    out = []
    for a in args.split(","):
        a = a.strip()
        m = re.match(r"^(.*?)(\w+)\s*(\[\d*\])?$", a)  # type, name, optional array suffix
        ctype, arr = m.group(1).strip(), m.group(3)
        if arr:
            ctype += " *"  # an array parameter is a pointer
        out.append(c_type_to_rust(ctype))
    return out


def parse_header():
    text = strip_c_comments(open(HEADER).read())
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    structs = {}
    for m in re.finditer(r"typedef\s+struct\s+(\w+)\s*\{(.*?)\}\s*(\w+)\s*;", text, flags=re.S):
        fields = []
        for decl in m.group(2).split(";"):
            decl = decl.strip()
            if not decl:
                continue
            fm = re.match(r"^(.*?)(\w+)\s*(\[\w+\])?$", decl)
            fields.append((fm.group(2), c_type_to_rust(fm.group(1).strip()) + (fm.group(3) or "")))
        structs[m.group(3)] = fields
    body = re.sub(r"typedef\s+(struct|enum)\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", text, flags=re.S)
    funcs = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(fr_\w+)\s*\(([^()]*)\)\s*;", body):
        ret = m.group(1).strip()
        funcs[m.group(2)] = (None if ret == "void" else c_type_to_rust(ret), parse_c_params(m.group(3)))
    return structs, funcs


def parse_shim():
    text = open(SHIM).read()
    text = re.sub(r"//.*$", "", text, flags=re.M)
    structs = {}
    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[[^\]]*\]\s*)*pub struct (\w+)\s*\{(.*?)\}", text, flags=re.S):
        fields = [(fm.group(1), " ".join(fm.group(2).split())) for fm in re.finditer(r"pub (\w+)\s*:\s*([^,]+),", m.group(2))]
        structs[m.group(1)] = fields
    ext = re.search(r'extern "C"\s*\{(.*?)\n\}', text, flags=re.S).group(1)
    funcs = {}
    for m in re.finditer(r"pub fn (\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+?))?\s*;", ext, flags=re.S):
        params = [" ".join(p.split(":", 1)[1].split()) for p in m.group(2).split(",") if p.strip()]
        funcs[m.group(1)] = (" ".join(m.group(3).split()) if m.group(3) else None, params)
    wrappers = set(re.findall(r"^pub fn (\w+)", text, flags=re.M))
    return structs, funcs, wrappers


def test_the_parsers_see_the_whole_header():
    structs, funcs = parse_header()
    assert {"fr_config", "fr_imaginary", "fr_rgb", "fr_render_opts", "fr_multi_stats"} <= set(structs)
    assert len(funcs) >= 56 and funcs["fr_render_rgb8"] == ("c_int", ["*const fr_config", "*mut u8", "usize"])
    assert funcs["fr_config_new"] == (None, ["*mut fr_config", "u32"])
    assert funcs["fr_last_error"] == ("*const c_char", [])
    assert funcs["fr_debug_sample_view"][1][-1] == "*mut f64"  # `double out[8]`
    # every function the header declares is exported by the ctypes binding too (tests/test_abi_cpu.py checks the .so)
    from fractal_renderer_amd import _native

    assert set(funcs) == set(_native.PROTOTYPES), set(funcs) ^ set(_native.PROTOTYPES)


def test_repr_c_structs_match_the_header_field_for_field():
    hs, _ = parse_header()
    rs, _, _ = parse_shim()
    assert {"fr_config", "fr_imaginary", "fr_rgb"} <= set(rs)
    for name, fields in rs.items():
        assert name in hs, "%s is not a struct of the header" % name
        assert fields == hs[name], (name, fields, hs[name])


def test_every_extern_matches_the_header_signature():
    _, hf = parse_header()
    _, rf, _ = parse_shim()
    assert len(rf) >= 25
    header_text = open(HEADER).read()
    shim_text = open(SHIM).read()
    assert re.search(r"#define FR_ABI_VERSION (\d+)", header_text).group(1) == re.search(
        r"pub const FR_ABI_VERSION: c_int = (\d+);", shim_text).group(1)
    for name, (ret, params) in rf.items():
        assert name in hf, "%s is declared by the shim but not by include/fractal_hip.h" % name
        assert (ret, params) == hf[name], (name, (ret, params), hf[name])


def test_the_reference_surface_has_externs_and_safe_wrappers():
    _, rf, wrappers = parse_shim()
    need_externs = {"fr_render_rgb8", "fr_render_rows_rgb8", "fr_render_rows_rgba8", "fr_pixel", "fr_recursive", "fr_escape_rows",
                    "fr_colour_rgb8", "fr_render_rgb8_multi", "fr_init_devices", "fr_render_fern_rgb8", "fr_last_error", "fr_config_new"}
    assert need_externs <= set(rf), need_externs - set(rf)
    # get_image (src/lib.rs:253), get_recursive_pixel (calc/src/lib.rs:199), recursive (:245), the GUI frame (src/gui.rs:71-72)
    assert {"render_into", "get_recursive_pixel", "recursive", "escape_rows", "colour_into", "render_rgba_into", "fern_into",
            "use_devices", "last_error"} <= wrappers, wrappers
    assert os.path.exists(os.path.join(ROOT, "rust", "gui.patch.rs")) and os.path.exists(os.path.join(ROOT, "rust", "get_image.patch.rs"))
