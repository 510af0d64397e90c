"""Pins held by the reference itself: the product's decoders and OBJ reader against fixtures written by the
reference's OWN ext/stb_image.h and ext/tiny_obj_loader.h (tests/golden/make_ref_fixtures.py runs them through
oracle/ref_dump.cpp in the build container; reference call sites src/scene.cpp:23-48, 471-527).

Bar: BIT FOR BIT. JPEG decoding is integer work (Huffman, integer IDCT, fixed-point upsampling and colour
conversion); RGBE decoding is ldexp on integers; OBJ parsing is digit arithmetic in double with one rounding to float,
and index bookkeeping. Whole images and corner lists are compared through SHA-256, with crops / the first and last
100 faces kept in the fixture so that a failure can be looked at.
"""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN
from tracy_amd import capi

ASSETS = os.path.join(capi.ASSET_ROOT, "data")
JPEGS = ["earth.jpg", "helmet_albedo.jpg", "helmet_emissive.jpg", "helmet_metal.jpg", "helmet_normal.jpg", "helmet_roughness.jpg"]
OBJS = ["teapot.obj", "stanford_dragon.obj", "stanford_bunny.obj", "helmet.obj"]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def decode_jpeg(path):
    lib = capi.scene_lib()
    w, h, p = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    rc = lib.tsc_decode_jpeg_file(path.encode(), C.byref(w), C.byref(h), C.byref(p))
    assert rc == 0, lib.tsc_last_error().decode()
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib.tsc_free_pixels(p)


def decode_hdr(path):
    lib = capi.scene_lib()
    w, h, p = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_float)()
    rc = lib.tsc_decode_hdr_file(path.encode(), C.byref(w), C.byref(h), C.byref(p))
    assert rc == 0, lib.tsc_last_error().decode()
    try:
        return np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    finally:
        lib.tsc_free_pixels(p)


def parse_obj(path):
    lib = capi.scene_lib()
    cnt = capi.tsc_obj_counts()
    v, vt, vn = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
    sizes, corners = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
    rc = lib.tsc_parse_obj_file(path.encode(), C.byref(cnt), C.byref(v), C.byref(vt), C.byref(vn), C.byref(sizes), C.byref(corners))
    assert rc == 0, lib.tsc_last_error().decode()

    def take(p, n, dt):
        a = np.ctypeslib.as_array(p, shape=(max(n, 1),))[:n].astype(dt, copy=True)
        lib.tsc_free_pixels(p)
        return a
    return (take(v, cnt.v_floats, np.float32), take(vt, cnt.vt_floats, np.float32), take(vn, cnt.vn_floats, np.float32),
            take(sizes, cnt.shapes, np.int32), take(corners, cnt.corners * 3, np.int32).reshape(-1, 3))


@pytest.mark.parametrize("name", JPEGS)
def test_jpeg_decoder_equals_stb_image_bit_for_bit(name):
    g = np.load(os.path.join(GOLDEN, "ref_stb_jpeg.npz"))
    key = name.replace(".", "_")
    path = os.path.join(ASSETS, name)
    assert sha(open(path, "rb").read()) == str(g[key + "_file_sha256"]), "the asset is not the file the reference ships"
    px = decode_jpeg(path)
    w, h, comp = g[key + "_whc"]
    assert px.shape == (h, w, 4)
    y0, x0 = h // 2 - 32, w // 2 - 32
    mid, corner = px[y0:y0 + 64, x0:x0 + 64], px[h - 24:, w - 24:]
    assert int((mid != g[key + "_mid"]).sum()) == 0, f"{name}: middle crop differs, max |d| = {np.abs(mid.astype(int) - g[key + '_mid']).max()}"
    assert int((corner != g[key + "_corner"]).sum()) == 0, f"{name}: bottom-right corner differs"
    assert sha(px.tobytes()) == str(g[key + "_sha256"]), f"{name}: whole image differs from stbi_load"
    if comp == 1:   # grey JPEG: stbi_load(.., 4) replicates the luma, alpha 255 (src/scene.cpp:39)
        assert (px[..., 0] == px[..., 1]).all() and (px[..., 0] == px[..., 2]).all()
    assert (px[..., 3] == 255).all()


@pytest.mark.parametrize("tag", ["rle", "flat"])
def test_hdr_decoder_equals_stbi_loadf_bit_for_bit(tag):
    g = np.load(os.path.join(GOLDEN, "ref_stb_hdr.npz"))
    path = os.path.join(GOLDEN, f"probe_{tag}.hdr")
    assert sha(open(path, "rb").read()) == str(g[tag + "_file_sha256"])
    px = decode_hdr(path)
    want = g[tag + "_rgba"]
    assert px.shape == want.shape
    assert int((px.view(np.uint32) != want.view(np.uint32)).sum()) == 0
    assert float(px[..., :3].max()) >= 1000.0 and (px[..., 3] == 1.0).all()


@pytest.mark.parametrize("name", OBJS)
def test_obj_reader_equals_tiny_obj_loader(name):
    """teapot.obj is all quads (tinyobj splits each along its shorter diagonal), helmet.obj has v/vt/vn corners, the
    dragon and the bunny are plain triangles without normals."""
    g = np.load(os.path.join(GOLDEN, "ref_tinyobj.npz"))
    key = name.replace(".", "_")
    path = os.path.join(ASSETS, name)
    assert sha(open(path, "rb").read()) == str(g[key + "_file_sha256"]), "the asset is not the file the reference ships"
    v, vt, vn, sizes, corners = parse_obj(path)
    nv, nvt, nvn, nshapes, ncorners = (int(x) for x in g[key + "_counts"])
    assert (v.size, vt.size, vn.size, sizes.size, corners.shape[0]) == (nv, nvt, nvn, nshapes, ncorners)
    assert sizes.tolist() == g[key + "_shape_sizes"].tolist()
    assert sha(v.tobytes()) == str(g[key + "_v_sha256"]), "positions differ from tinyobj's parse"
    assert sha(vt.tobytes()) == str(g[key + "_vt_sha256"])
    assert sha(vn.tobytes()) == str(g[key + "_vn_sha256"])
    assert int((corners[:300] != g[key + "_first"]).sum()) == 0, "first 100 faces differ"
    assert int((corners[-300:] != g[key + "_last"]).sum()) == 0, "last 100 faces differ"
    assert sha(corners.tobytes()) == str(g[key + "_corners_sha256"]), "corner triples differ from tinyobj's"


def test_obj_reader_rejects_out_of_range_and_zero_indices(tmp_path):
    """tinyobj fails the load on index 0 and lets out-of-range corners through (Scene::Init then reads out of bounds);
    this reader fails both with a message -- never an out-of-bounds read."""
    lib = capi.scene_lib()
    cnt = capi.tsc_obj_counts()
    head = "v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nvt 0 0\n"
    for body, what in (("f 1 2 0\n", "index 0"), ("f 1 2 4\n", "outside"), ("f 1 2 -4\n", "outside"), ("f 1/2 2/1 3/1\n", "outside"),
                       ("f 1//2 2//1 3//1\n", "outside"), ("f 1 2 x\n", "index 0"), ("f 3 2 5\nv 5 5 5\nv 6 6 6\n", "outside")):
        p = tmp_path / "bad.obj"
        p.write_text(head + body)
        rc = lib.tsc_parse_obj_file(str(p).encode(), C.byref(cnt), None, None, None, None, None)
        assert rc != 0 and what in lib.tsc_last_error().decode(), (body, lib.tsc_last_error().decode())
    good = tmp_path / "good.obj"
    good.write_text(head + "f -3 -2 -1\nf 1/1/1 2/1/1 3/1/1\n")
    assert lib.tsc_parse_obj_file(str(good).encode(), C.byref(cnt), None, None, None, None, None) == 0
    assert (cnt.shapes, cnt.corners) == (1, 6)


def test_obj_polygons_follow_tinyobj(tmp_path):
    """Quads split along the shorter diagonal (tiny_obj_loader.h:1457-1565), a convex pentagon by ear clipping from
    corner 0 (:1566-1852), groups make shapes."""
    src = ("v 0 0 0\nv 4 0 0\nv 4 1 0\nv 0 1 0\n"          # 1-4: a 4x1 rectangle, both diagonals equal -> [0,1,3][1,2,3]
           "v 0 0 1\nv 1 0 1\nv 3 1 1\nv 0 1 1\n"          # 5-8: |v7-v5|^2 = 10 > |v8-v6|^2 = 2 -> [0,1,3][1,2,3]
           "v 0 0 2\nv 3 0 2\nv 1 1 2\nv 0 1 2\n"          # 9-12: |v11-v9|^2 = 2 < |v12-v10|^2 = 10 -> [0,1,2][0,2,3]
           "v 0 0 3\nv 2 0 3\nv 3 1 3\nv 1 2 3\nv -1 1 3\n"  # 13-17: convex pentagon
           "f 1 2 3 4\ng second\nf 5 6 7 8\nf 9 10 11 12\no third\nf 13 14 15 16 17\n")
    p = tmp_path / "poly.obj"
    p.write_text(src)
    _, _, _, sizes, corners = parse_obj(str(p))
    assert sizes.tolist() == [6, 12, 9]
    vi = corners[:, 0].tolist()
    assert vi[0:6] == [0, 1, 3, 1, 2, 3]
    assert vi[6:12] == [4, 5, 7, 5, 6, 7]
    assert vi[12:18] == [8, 9, 10, 8, 10, 11]
    assert vi[18:27] == [12, 13, 14, 12, 14, 15, 12, 15, 16]
    assert (corners[:, 1:] == -1).all()
