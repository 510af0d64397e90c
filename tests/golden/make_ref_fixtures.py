"""Fixtures made by the REFERENCE'S OWN third-party loaders (build container only).

    make -C oracle ref && python tests/golden/make_ref_fixtures.py

oracle/_ref/ref_dump is oracle/ref_dump.cpp compiled against /root/reference/ext/stb_image.h and
/root/reference/ext/tiny_obj_loader.h as they stand (no copies, no stand-ins) and calls them the way
Scene::Init does (src/scene.cpp:23-48, 471-476). This script freezes what they return for the files the
reference ships, as small data fixtures:

  tests/golden/ref_stb_jpeg.npz   per shipped JPEG: width, height, channels, SHA-256 of stbi_load(.., 4)'s RGBA8
                                  bytes, a 64x64 crop from the middle and the 24x24 bottom-right corner (the edge
                                  rules of the chroma upsampling live there), SHA-256 of the file itself
  tests/golden/ref_stb_hdr.npz    two synthetic Radiance files written by this script (run-length coded scanlines;
                                  a flat file too narrow for RLE) with stbi_loadf(.., 4)'s floats, all of them
  tests/golden/ref_tinyobj.npz    per shipped OBJ: array lengths, SHA-256 of attrib.vertices / texcoords / normals,
                                  corners per shape, SHA-256 of all (v, vt, vn) corner triples, the first and the last
                                  300 corners (100 faces)

The GPU box has neither the reference nor ref_dump: only these fixtures travel (tests/test_reference_pins.py).
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
REF_DATA = "/root/reference/data"

JPEGS = ["earth.jpg", "helmet_albedo.jpg", "helmet_emissive.jpg", "helmet_metal.jpg", "helmet_normal.jpg", "helmet_roughness.jpg"]
OBJS = ["teapot.obj", "stanford_dragon.obj", "stanford_bunny.obj", "helmet.obj"]


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def run_dump(mode, path):
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as t:
        out = t.name
    try:
        subprocess.run([DUMP, mode, path, out], check=True)
        return open(out, "rb").read()
    finally:
        os.unlink(out)


def rgbe(r, g, b):
    """float -> RGBE as Radiance writes it (frexp of the largest component)"""
    m = max(r, g, b)
    if m < 1e-32:
        return (0, 0, 0, 0)
    import math
    f, e = math.frexp(m)
    s = f * 256.0 / m
    return (int(r * s), int(g * s), int(b * s), e + 128)


def write_hdr(path, w, h, rle):
    """A small deterministic Radiance picture: smooth gradients (long RLE runs in the exponent plane), a bright spot,
    a black run and a noisy band (literal runs)."""
    rows = []
    for y in range(h):
        row = []
        for x in range(w):
            r = 0.05 + 0.9 * x / max(1, w - 1)
            g = 0.5 * (1 + np.sin(0.7 * x + 1.3 * y)) * (4.0 if (x // 3) % 2 else 0.02)
            b = 1000.0 if (x, y) == (w // 2, h // 2) else (0.0 if x < 3 else 0.25)
            if y == h - 1:
                r = g = b = 0.0
            row.append(rgbe(float(r), float(g), float(b)))
        rows.append(row)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n")
        f.write(f"-Y {h} +X {w}\n".encode())
        for row in rows:
            if not rle:
                for px in row:
                    f.write(bytes(px))
                continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                plane = [px[c] for px in row]
                i = 0
                while i < w:
                    run = 1
                    while i + run < w and run < 127 and plane[i + run] == plane[i]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, plane[i]]))
                        i += run
                    else:
                        j = i
                        while j < w and j - i < 128:
                            if j + 2 < w and plane[j] == plane[j + 1] == plane[j + 2]:
                                break
                            j += 1
                        f.write(bytes([j - i]) + bytes(plane[i:j]))
                        i = j


def main():
    if not os.path.exists(DUMP):
        sys.exit("build oracle/_ref/ref_dump first: make -C oracle ref")
    # ---- stb_image, JPEG
    jp = {}
    for name in JPEGS:
        path = os.path.join(REF_DATA, name)
        raw = run_dump("jpeg", path)
        w, h, bpp = struct.unpack("<3i", raw[:12])
        px = np.frombuffer(raw[12:], dtype=np.uint8).reshape(h, w, 4)
        key = name.replace(".", "_")
        jp[key + "_whc"] = np.array([w, h, bpp], dtype=np.int32)
        jp[key + "_sha256"] = np.array(sha(px.tobytes()))
        jp[key + "_file_sha256"] = np.array(sha(open(path, "rb").read()))
        y0, x0 = h // 2 - 32, w // 2 - 32
        jp[key + "_mid"] = px[y0:y0 + 64, x0:x0 + 64].copy()
        jp[key + "_corner"] = px[h - 24:, w - 24:].copy()
        print(name, w, h, bpp, jp[key + "_sha256"])
    np.savez_compressed(os.path.join(HERE, "ref_stb_jpeg.npz"), **jp)
    # ---- stb_image, Radiance .hdr
    hd = {}
    for tag, w, h, rle in (("rle", 40, 12, True), ("flat", 5, 3, False)):
        path = os.path.join(HERE, f"probe_{tag}.hdr")
        write_hdr(path, w, h, rle)
        raw = run_dump("hdr", path)
        ww, hh, bpp = struct.unpack("<3i", raw[:12])
        assert (ww, hh) == (w, h)
        hd[tag + "_whc"] = np.array([ww, hh, bpp], dtype=np.int32)
        hd[tag + "_rgba"] = np.frombuffer(raw[12:], dtype=np.float32).reshape(h, w, 4).copy()
        hd[tag + "_file_sha256"] = np.array(sha(open(path, "rb").read()))
        print(path, ww, hh, bpp, float(hd[tag + "_rgba"].max()))
    np.savez_compressed(os.path.join(HERE, "ref_stb_hdr.npz"), **hd)
    # ---- tiny_obj_loader
    ob = {}
    for name in OBJS:
        path = os.path.join(REF_DATA, name)
        raw = run_dump("obj", path)
        nv, nvt, nvn, nshapes = struct.unpack("<4i", raw[:16])
        off = 16
        v = np.frombuffer(raw, dtype=np.float32, count=nv, offset=off); off += nv * 4
        vt = np.frombuffer(raw, dtype=np.float32, count=nvt, offset=off); off += nvt * 4
        vn = np.frombuffer(raw, dtype=np.float32, count=nvn, offset=off); off += nvn * 4
        sizes, corners = [], []
        for _ in range(nshapes):
            (n,) = struct.unpack_from("<i", raw, off); off += 4
            corners.append(np.frombuffer(raw, dtype=np.int32, count=n * 3, offset=off)); off += n * 12
            sizes.append(n)
        assert off == len(raw)
        allc = np.concatenate(corners) if corners else np.zeros(0, np.int32)
        key = name.replace(".", "_")
        ob[key + "_counts"] = np.array([nv, nvt, nvn, nshapes, allc.size // 3], dtype=np.int64)
        ob[key + "_v_sha256"] = np.array(sha(v.tobytes()))
        ob[key + "_vt_sha256"] = np.array(sha(vt.tobytes()))
        ob[key + "_vn_sha256"] = np.array(sha(vn.tobytes()))
        ob[key + "_shape_sizes"] = np.array(sizes, dtype=np.int32)
        ob[key + "_corners_sha256"] = np.array(sha(allc.tobytes()))
        ob[key + "_first"] = allc[:900].reshape(-1, 3).copy()
        ob[key + "_last"] = allc[-900:].reshape(-1, 3).copy()
        ob[key + "_file_sha256"] = np.array(sha(open(path, "rb").read()))
        print(name, "floats", nv, nvt, nvn, "shapes", sizes, "corners", allc.size // 3)
    np.savez_compressed(os.path.join(HERE, "ref_tinyobj.npz"), **ob)


if __name__ == "__main__":
    main()
