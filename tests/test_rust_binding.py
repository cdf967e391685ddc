"""The Rust binding shown in INTEGRATION.md (the `extern "C"` block a maintainer of the reference pastes into src/gpu.rs) checked
mechanically against include/rt_amd.h: no Rust toolchain exists here, so nothing compiles that block — one transposed argument would
break the drop-in for src/main.rs:811-832 silently.  Every function of the block must exist in the header with the same arity and,
argument by argument, the same type class (c_double <-> double, *const c_double <-> const double* / const double[3], u32 <-> uint32_t,
usize <-> size_t, *mut RtScene <-> rt_scene*, ...), the same return type, and `#[repr(C)] RtCamera` must list rt_camera's fields in
rt_camera's order with its array lengths.  The test also proves that it bites: a copy of the block with two arguments swapped fails."""
import os
import re

from conftest import ROOT

C_SCALARS = {"double": "f64", "int": "i32", "int32_t": "i32", "uint32_t": "u32", "uint64_t": "u64", "size_t": "usize", "uint8_t": "u8", "char": "char",
             "float": "f32", "void": "void", "rt_scene": "Scene", "rt_rng": "Rng", "rt_camera": "Camera", "long long": "i64",
             "unsigned long long": "u64"}
RUST_SCALARS = {"c_double": "f64", "c_int": "i32", "i32": "i32", "f32": "f32", "u32": "u32", "u64": "u64", "usize": "usize", "u8": "u8", "c_char": "char",
                "c_float": "f32", "c_void": "void", "RtScene": "Scene", "RtRng": "Rng", "RtCamera": "Camera", "i64": "i64"}


def _c_type(decl):
    """'const double rgb[3]' -> ('ptr_const', 'f64'); 'rt_scene*' -> ('ptr_mut', 'Scene'); 'uint32_t W' -> ('val', 'u32')."""
    d = decl.strip()
    arr = bool(re.search(r"\[\w*\]\s*$", d))
    d = re.sub(r"\[\w*\]\s*$", "", d).strip()
    const = bool(re.match(r"const\b", d))
    d = re.sub(r"^const\s+", "", d)
    stars = d.count("*")
    d = d.replace("*", " ").strip()
    words = d.split()
    # the base type is the longest prefix that names a known type; what follows (if anything) is the parameter's name
    base = None
    for k in range(len(words), 0, -1):
        if " ".join(words[:k]) in C_SCALARS:
            base = C_SCALARS[" ".join(words[:k])]
            break
    assert base is not None, f"unknown C type in {decl!r}"
    depth = stars + (1 if arr else 0)
    if depth == 0:
        return ("val", base)
    kind = ("ptr_const", base) if const else ("ptr_mut", base)
    for _ in range(depth - 1):
        kind = ("ptr_mut", kind)
    return kind


def _rust_type(t):
    t = t.strip()
    m = re.match(r"\*(const|mut)\s+(.*)$", t)
    if m:
        return ("ptr_const" if m.group(1) == "const" else "ptr_mut", _rust_type(m.group(2)) if m.group(2).startswith("*") else RUST_SCALARS[m.group(2).strip()])
    return ("val", RUST_SCALARS[t])


def c_prototypes(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    text = re.sub(r"^\s*#.*$", "", text, flags=re.M)
    text = re.sub(r"typedef struct rt_camera \{.*?\} rt_camera;", "", text, flags=re.S)
    text = re.sub(r"enum\s*\w*\s*\{.*?\};", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"([\w\s\*]+?)\b(rt_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        arg_types = [] if args in ("", "void") else [_c_type(a) for a in args.split(",")]
        protos[name] = (_c_type(ret + " x") if ret != "void" else ("val", "void"), arg_types)
    return protos


def rust_prototypes(block):
    block = re.sub(r"//[^\n]*", "", block)
    protos = {}
    for m in re.finditer(r"pub fn (rt_[a-z0-9_]+)\s*\(([^()]*)\)\s*(->\s*([^;]+))?;", block, flags=re.S):
        name, args, ret = m.group(1), m.group(2).strip(), m.group(4)
        arg_types = [] if not args else [_rust_type(a.split(":", 1)[1]) for a in args.split(",") if a.strip()]
        protos[name] = (_rust_type(ret) if ret else ("val", "void"), arg_types)
    return protos


def camera_fields_c(text):
    body = re.search(r"typedef struct rt_camera \{(.*?)\} rt_camera;", text, flags=re.S).group(1)
    out = []
    for stmt in body.split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        assert stmt.startswith("double")
        for f in stmt[len("double"):].split(","):
            m = re.match(r"\s*(\w+)\s*(\[(\d+)\])?\s*$", f)
            out.append((m.group(1), int(m.group(3)) if m.group(3) else 1))
    return out


def camera_fields_rust(block):
    body = re.search(r"#\[repr\(C\)\]\s*pub struct RtCamera \{(.*?)\n\}", block, flags=re.S).group(1)
    body = re.sub(r"//[^\n]*", "", body)
    out = []
    for m in re.finditer(r"pub (\w+):\s*(\[c_double;\s*(\d+)\]|c_double)", body):
        out.append((m.group(1), int(m.group(3)) if m.group(3) else 1))
    return out


def _load():
    hdr = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    rust = re.search(r"```rust\n(use std::os::raw.*?)```", md, flags=re.S).group(1)
    return hdr, rust


def compare(hdr, rust):
    cp, rp = c_prototypes(hdr), rust_prototypes(rust)
    problems = []
    for name, (rret, rargs) in rp.items():
        if name not in cp:
            problems.append(f"{name}: not declared in include/rt_amd.h")
            continue
        cret, cargs = cp[name]
        if cret != rret:
            problems.append(f"{name}: return type {rret} vs header {cret}")
        if len(cargs) != len(rargs):
            problems.append(f"{name}: {len(rargs)} arguments vs header {len(cargs)}")
            continue
        for k, (a, b) in enumerate(zip(rargs, cargs)):
            if a != b:
                problems.append(f"{name}: argument {k} is {a} vs header {b}")
    if camera_fields_rust(rust) != camera_fields_c(hdr):
        problems.append(f"RtCamera fields {camera_fields_rust(rust)} vs rt_camera {camera_fields_c(hdr)}")
    return problems, cp, rp


def test_rust_extern_block_matches_the_c_header():
    hdr, rust = _load()
    problems, cp, rp = compare(hdr, rust)
    assert len(rp) == len(re.findall(r"pub fn rt_", rust)) >= 40                       # every line of the block was understood
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)))
    assert set(cp) == declared and len(cp) >= 60                                        # ... and every prototype of the header
    assert problems == []
    # everything the drop-in for src/main.rs:767-835 needs is bound
    for need in ("rt_scene_create", "rt_render", "rt_render_multi", "rt_write_ppm", "rt_format_color", "rt_sphere", "rt_aarect", "rt_cube",
                 "rt_translate", "rt_rotate", "rt_bvh", "rt_constant_medium", "rt_lights_push", "rt_scene_set_world", "rt_last_error"):
        assert need in rp


def test_the_check_fails_on_a_swapped_or_retyped_argument():
    hdr, rust = _load()
    # rt_moving_sphere(.., t0: c_double, t1: c_double, r: c_double, mat: c_int): swap the last two
    bad = rust.replace("t1: c_double, r: c_double, mat: c_int) -> c_int;", "t1: c_double, mat: c_int, r: c_double) -> c_int;")
    assert bad != rust and any("rt_moving_sphere" in p for p in compare(hdr, bad)[0])
    # rt_render's frame size as u64 instead of u32
    bad = rust.replace("w: u32, h: u32,\n                     spp: u32", "w: u64, h: u32,\n                     spp: u32")
    assert bad != rust and any(p.startswith("rt_render:") for p in compare(hdr, bad)[0])
    # a dropped argument, a const that became mut, a camera field out of order
    bad = rust.replace("pub fn rt_rotate(s: *mut RtScene, axis: c_int, h: c_int, angle: c_double)", "pub fn rt_rotate(s: *mut RtScene, h: c_int, angle: c_double)")
    assert any("rt_rotate" in p for p in compare(hdr, bad)[0])
    bad = rust.replace("pub fn rt_cube(s: *mut RtScene, min: *const c_double", "pub fn rt_cube(s: *mut RtScene, min: *mut c_double")
    assert any("rt_cube" in p for p in compare(hdr, bad)[0])
    bad = rust.replace("pub vfov: c_double, pub aspect: c_double", "pub aspect: c_double, pub vfov: c_double")
    assert any("RtCamera" in p for p in compare(hdr, bad)[0])
