"""Malformed-input behaviour of the host-side asset ingest (csrc/rt_jpeg.cpp, csrc/rt_obj.cpp): the bytes come from files a user
hands to the renderer (`image::open`, src/main.rs:248,491; `tobj::load_obj`, src/mesh.rs:40), so every malformed input must end
in an error value — never a crash, an out-of-range read or an exception crossing the C boundary.  These tests also run against the
AddressSanitizer/UBSan build of the host objects (`make -C raytracinginrust_amd/csrc asan`, tests/test_sanitizers.py)."""
import ctypes as C
import random

import numpy as np
import pytest

from conftest import golden_path

from raytracinginrust_amd import scenes


def _decode(lib, data: bytes):
    lib.rt_decode_jpeg_rgb8.restype = C.c_void_p
    lib.rt_decode_jpeg_rgb8.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.rt_free.argtypes = [C.c_void_p]
    w, h = C.c_uint32(), C.c_uint32()
    ptr = lib.rt_decode_jpeg_rgb8(data, len(data), C.byref(w), C.byref(h))
    if not ptr:
        return None
    out = C.string_at(ptr, 3 * w.value * h.value)
    lib.rt_free(ptr)
    return out, w.value, h.value


def _parse_obj(lib, data: bytes, offset=(0.0, 0.0, 0.0), scale=1.0):
    lib.rt_parse_obj.restype = C.c_int
    lib.rt_parse_obj.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_double), C.c_double, C.POINTER(C.POINTER(C.c_double)),
                                 C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32)]
    lib.rt_free.argtypes = [C.c_void_p]
    pos, idx = C.POINTER(C.c_double)(), C.POINTER(C.c_uint32)()
    n_pos, n_idx = C.c_uint32(), C.c_uint32()
    rc = lib.rt_parse_obj(data, len(data), (C.c_double * 3)(*offset), scale, C.byref(pos), C.byref(n_pos), C.byref(idx), C.byref(n_idx))
    if rc != 0:
        return None
    p = np.ctypeslib.as_array(pos, shape=(n_pos.value * 3,)).copy().reshape(-1, 3) if n_pos.value else np.zeros((0, 3))
    i = np.ctypeslib.as_array(idx, shape=(n_idx.value,)).copy() if n_idx.value else np.zeros(0, dtype=np.uint32)
    lib.rt_free(pos); lib.rt_free(idx)
    return p, i


def test_obj_ingest_agrees_with_the_python_loader(pbe):
    """csrc/rt_obj.cpp (what the C++ host's Mesh::load_obj calls) vs scenes.load_obj (an independent implementation) on the
    reference's teapot.obj, with a scale and an offset (src/mesh.rs:51)."""
    path = scenes.asset_path("teapot.obj")
    off, sc = (268.0, 340.0, 258.0), 1.5
    pos, idx = scenes.load_obj(path, off, sc)
    got = _parse_obj(pbe.lib, open(path, "rb").read(), off, sc)
    assert got is not None
    assert np.array_equal(got[0], np.asarray(pos, dtype=np.float64)) and np.array_equal(got[1], np.asarray(idx, dtype=np.uint32))


def test_obj_ingest_semantics(pbe):
    lib = pbe.lib
    quad = b"v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nf 1 2 3 4\n"
    p, i = _parse_obj(lib, quad)
    assert p.shape == (4, 3) and i.tolist() == [0, 1, 2, 0, 2, 3]                      # fan triangulation
    p, i = _parse_obj(lib, b"v 0 0 0\nv 1 0 0\nv 1 1 0\nf -3/1/1 -2//2 -1\n")
    assert i.tolist() == [0, 1, 2]                                                      # relative indices, v/vt/vn forms
    p, i = _parse_obj(lib, quad + b"o second\nv 5 5 5\nf 1 2 5\n")
    assert p.shape == (4, 3) and len(i) == 6                                            # models[0] only
    p, i = _parse_obj(lib, b"# comment\r\nv 0.1 0.2 0.3\r\nv 1 0 0\r\nv 0 1 0\r\nf 1 2 3\r\n")
    assert p[0].tolist() == [float(np.float32(0.1)), float(np.float32(0.2)), float(np.float32(0.3))]   # f32, then widened
    p, i = _parse_obj(lib, b"v 0 0 0\nv 1 0 0\nv 1 1 0\nf 1 2 3")                        # no trailing newline
    assert i.tolist() == [0, 1, 2]
    p, i = _parse_obj(lib, b"\x00\xff\xfe garbage \n\n\nf\n")        # an element-less face: an empty model, not an error
    assert len(p) == 0 and len(i) == 0


@pytest.mark.parametrize("data,msg", [
    (b"f 1 2 3\nv 0 0 0\nv 1 0 0\nv 0 1 0\n", "out of bounds"),          # `f` before any `v`
    (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 4\n", "out of bounds"),          # oversized index
    (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 -4\n", "out of bounds"),         # negative beyond the start
    (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n", "out of bounds"),          # OBJ indices are 1-based
    (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 99999999999999999999\n", "face parse"),
    (b"v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 x 3\n", "face parse"),
    (b"v 0 0\nf 1 1 1\n", "position parse"),
    (b"v 0 zero 0\n", "position parse"),
    (b"", "no model"),
    (b"v 0 0 0\nv 1 0 0\nv 0 1 0\n", "no model"),
])
def test_obj_ingest_rejects_malformed_files(pbe, data, msg):
    assert _parse_obj(pbe.lib, data) is None
    import re
    assert re.search(msg, pbe.lib.rt_last_error().decode())


def test_mesh_load_obj_errors_do_not_crash(pbe, tmp_path):
    from raytracinginrust_amd.api import SceneBuilder
    b = SceneBuilder(pbe)
    m = b.Lambertian(b.ConstantTexture((1, 1, 1)))
    lib = pbe.lib
    lib.rt_mesh_load_obj.restype = C.c_int
    lib.rt_mesh_load_obj.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.c_double, C.c_int]
    off = (C.c_double * 3)(0, 0, 0)
    assert lib.rt_mesh_load_obj(b.h, str(tmp_path / "missing.obj").encode(), off, 1.0, m.id) < 0          # the reference: Err -> unwrap panic
    assert b"Failed to load obj file" in lib.rt_last_error()
    bad = tmp_path / "bad.obj"; bad.write_bytes(b"v 0 0 0\nf 1 2 3\n")
    assert lib.rt_mesh_load_obj(b.h, str(bad).encode(), off, 1.0, m.id) < 0
    assert lib.rt_mesh_load_obj(b.h, scenes.asset_path("teapot.obj").encode(), off, 1.0, 12345) < 0         # bad material handle
    h = lib.rt_mesh_load_obj(b.h, scenes.asset_path("teapot.obj").encode(), off, 1.0, m.id)
    assert h >= 0


def test_obj_ingest_mutation_fuzz(pbe):
    """Seeded byte mutations of the real mesh file: every outcome is either a mesh whose indices are in range or an error."""
    src = open(scenes.asset_path("teapot.obj"), "rb").read()
    head = src[:6000]                      # a few hundred `v` lines ... (keeps each case cheap)
    tail = src[-3000:]                     # ... and some `f` lines referring to vertices mostly outside that window
    rnd = random.Random(0xB0B)
    alphabet = b"0123456789-/. \nvfog#e+\x00\xff"
    ok = bad = 0
    for case in range(400):
        data = bytearray(head + tail if case % 2 else src[: rnd.randrange(1, len(src))])
        for _ in range(rnd.randrange(1, 12)):
            k = rnd.randrange(len(data))
            r = rnd.random()
            if r < 0.5:
                data[k] = rnd.choice(alphabet)
            elif r < 0.75:
                del data[k: k + rnd.randrange(1, 40)]
            else:
                data[k:k] = bytes(rnd.choice(alphabet) for _ in range(rnd.randrange(1, 20)))
        got = _parse_obj(pbe.lib, bytes(data))
        if got is None:
            bad += 1
        else:
            ok += 1
            p, i = got
            assert len(i) % 3 == 0 and (len(i) == 0 or int(i.max()) < len(p))
    assert ok + bad == 400 and bad > 0


def _jpeg_cases():
    """Hand-made hostile streams (the classes ADVICE.md / VERDICT.md name) + seeded mutations of real files."""
    real = open(golden_path("earthmap_256x128_444.jpg"), "rb").read()
    grey = open(golden_path("earthmap_256x128_grey.jpg"), "rb").read()
    cases = [b"", b"\xff", b"\xff\xd8", b"\xff\xd8\xff", b"\xff\xd8\xff\xff\xff\xff", b"\xff\xd8" + b"\xff" * 64,
             b"\xff\xd8\xff\xd9", b"\xff\xd8\xff\xe0\x00", b"\xff\xd8\xff\xe0\xff\xff" + b"\x00" * 10]
    # truncation at every marker boundary and at a spread of other offsets
    for k in list(range(2, 700, 7)) + list(range(700, len(real), 997)):
        cases.append(real[:k])
    # SOF with hostile dimensions (65535 x 65535) and zero dimensions
    sof = real.index(b"\xff\xc0")
    for hw in (b"\xff\xff\xff\xff", b"\x00\x00\x00\x10", b"\x00\x10\x00\x00", b"\x40\x00\x40\x00"):
        cases.append(real[: sof + 5] + hw + real[sof + 9:])
    # 16-bit quantisation table flag (pq = 1) and table id 7
    dqt = real.index(b"\xff\xdb")
    cases.append(real[: dqt + 4] + bytes([0x10 | real[dqt + 4]]) + real[dqt + 5:])
    cases.append(real[: dqt + 4] + bytes([0x07]) + real[dqt + 5:])
    # Huffman tables: oversubscribed counts, counts summing past 256, bad class / id
    dht = real.index(b"\xff\xc4")
    cases.append(real[: dht + 5] + b"\xff" * 16 + real[dht + 21:])
    cases.append(real[: dht + 5] + b"\x00" * 16 + real[dht + 21:])
    cases.append(real[: dht + 4] + b"\x25" + real[dht + 5:])
    # scan header naming missing tables / wrong component count
    sos = real.index(b"\xff\xda")
    cases.append(real[: sos + 4] + b"\x02" + real[sos + 5:])
    cases.append(real[: sos + 6] + b"\x33" + real[sos + 7:])
    # entropy data replaced by 0xFF / zeros / ones (runs DC predictors and run lengths out of range)
    body = sos + 14
    for fill in (b"\x00", b"\xff\x00", b"\xaa", b"\x7f"):
        cases.append(real[:body] + fill * 4000 + b"\xff\xd9")
    cases.append(real[:body])                                   # scan header, no data at all
    rnd = random.Random(0x5EED)
    for src in (real, grey):
        for _ in range(150):
            data = bytearray(src)
            hdr = rnd.random() < 0.6
            for _ in range(rnd.randrange(1, 8)):
                k = rnd.randrange(2, 700 if hdr else len(data))
                data[k] = rnd.randrange(256)
            cases.append(bytes(data))
    return cases


def test_jpeg_ingest_survives_malformed_streams(pbe):
    n_ok = n_err = 0
    for data in _jpeg_cases():
        got = _decode(pbe.lib, data)
        if got is None:
            n_err += 1
            assert pbe.lib.rt_last_error()
        else:
            n_ok += 1
            rgb, w, h = got
            assert 0 < w * h <= 64 << 20 and len(rgb) == 3 * w * h
    assert n_err > 100 and n_ok > 0          # mutations in the entropy data still decode (to different pixels); header damage must not
