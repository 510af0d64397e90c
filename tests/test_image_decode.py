"""Texture decoders of the standalone scene loader: baseline JPEG and Radiance
RGBE .hdr, csrc/host/image_decode.cpp, through the C ABI of include/tracy_scene.h.

Parity unpinned against stb_image (the reference holds no decoded-pixel fixtures). Pins used here:
  * JPEG: libjpeg-turbo through PIL on the shipped textures and on files PIL writes with every
    chroma layout / restart intervals / odd sizes. Both decoders use an LLM integer IDCT and a
    triangle chroma filter but round differently: max |d| <= 4 levels, mean |d| <= 0.15 at quality 90+;
    grey-scale images (no chroma path) max |d| <= 1.
  * HDR: hand-made RGBE files (flat and run-length encoded) whose texels are exact powers of two
    times 8-bit mantissas: compared exactly.
"""
import ctypes as C
import glob
import io
import os

import numpy as np
import pytest

from tracy_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def decode_jpeg(path):
    L = capi.scene_lib()
    w, h, p = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_uint8)()
    rc = L.tsc_decode_jpeg_file(str(path).encode(), C.byref(w), C.byref(h), C.byref(p))
    if rc:
        raise RuntimeError(L.tsc_last_error().decode())
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    L.tsc_free_pixels(p)
    return a


def decode_hdr(path):
    L = capi.scene_lib()
    w, h, p = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_float)()
    rc = L.tsc_decode_hdr_file(str(path).encode(), C.byref(w), C.byref(h), C.byref(p))
    if rc:
        raise RuntimeError(L.tsc_last_error().decode())
    a = np.ctypeslib.as_array(p, shape=(h.value, w.value, 4)).copy()
    L.tsc_free_pixels(p)
    return a


def pil_rgba(path):
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGBA"))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(ROOT, "assets", "data", "*.jpg"))), ids=os.path.basename)
def test_shipped_textures_against_libjpeg(path):
    a, b = decode_jpeg(path), pil_rgba(path)
    assert a.shape == b.shape and (a[..., 3] == 255).all()
    d = np.abs(a.astype(int) - b.astype(int))
    # (secondary cross-check against an independent decoder; the primary pin is bit-for-bit equality with the reference's
    # own stb_image, tests/test_reference_pins.py) grey pictures have no chroma up-sampling or colour conversion to
    # differ in: only the two IDCTs' rounding, 1 level
    grey = bool((a[..., 0] == a[..., 1]).all() and (a[..., 1] == a[..., 2]).all())
    assert grey == (("metal" in path) or ("roughness" in path))
    assert d.max() <= (1 if grey else 4)
    assert d.mean() <= 0.15


def synthetic_picture(w, h, seed):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 7.0 + seed), 127 + 120 * np.cos(y / 5.0), (x * 3 + y * 5) % 256], axis=-1)
    img += rng.normal(0, 6, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("size", [(64, 48), (37, 23), (8, 8), (1, 1), (17, 130)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])          # 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2)
def test_chroma_layouts_and_ragged_sizes(tmp_path, size, subsampling):
    from PIL import Image
    w, h = size
    f = tmp_path / "t.jpg"
    Image.fromarray(synthetic_picture(w, h, 3)).save(f, quality=92, subsampling=subsampling)
    a, b = decode_jpeg(f), pil_rgba(f)
    assert a.shape == (h, w, 4)
    d = np.abs(a.astype(int) - b.astype(int))
    if subsampling == 1 and w > 2:
        # 4:2:2: the decoder follows stb_image's h2v1 filter, whose second-to-last output sample is
        # 3*in[w-2] + in[w-1] (ext/stb_image.h:3462) where libjpeg weighs the nearer sample in[w-1] by 3
        d[:, 2 * ((w + 1) // 2) - 2] = 0
    assert d.max() <= 4 and d.mean() <= 0.6


def test_grey_restart_intervals_and_low_quality(tmp_path):
    from PIL import Image
    pic = synthetic_picture(100, 75, 5)
    f = tmp_path / "g.jpg"
    Image.fromarray(pic[..., 0]).save(f, quality=35)
    a, b = decode_jpeg(f), pil_rgba(f)
    assert (a[..., 0] == a[..., 1]).all() and (a[..., 0] == a[..., 2]).all()
    assert np.abs(a.astype(int) - b.astype(int)).max() <= 1
    f2 = tmp_path / "r.jpg"
    Image.fromarray(pic).save(f2, quality=90, subsampling=2, restart_marker_blocks=3)
    assert b"\xff\xdd" in f2.read_bytes()          # the file really carries a DRI segment
    a, b = decode_jpeg(f2), pil_rgba(f2)
    d = np.abs(a.astype(int) - b.astype(int))
    assert d.max() <= 4 and d.mean() <= 0.6


def test_progressive_and_garbage_are_refused(tmp_path):
    from PIL import Image
    f = tmp_path / "p.jpg"
    Image.fromarray(synthetic_picture(32, 32, 1)).save(f, progressive=True)
    with pytest.raises(RuntimeError, match="progressive"):
        decode_jpeg(f)
    g = tmp_path / "x.jpg"
    g.write_bytes(b"not a jpeg at all")
    with pytest.raises(RuntimeError, match="SOI"):
        decode_jpeg(g)
    t = tmp_path / "t.jpg"
    Image.fromarray(synthetic_picture(64, 64, 2)).save(t, quality=90)
    t.write_bytes(t.read_bytes()[:300])            # truncated inside the tables / scan
    with pytest.raises(RuntimeError):
        decode_jpeg(t)
    with pytest.raises(RuntimeError, match="cannot read"):
        decode_jpeg(tmp_path / "missing.jpg")


# ------------------------------------------------------------------------------- Radiance HDR
def rgbe_pixels(w, h, seed):
    rng = np.random.default_rng(seed)
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    px[..., 3] = rng.integers(120, 140, size=(h, w), dtype=np.uint8)
    px[0, 0, 3] = 0                                  # exponent 0 decodes to black
    px[h // 2, :, :] = px[h // 2, 0, :]              # a constant scanline: long runs for the RLE writer
    return px


def rgbe_expected(px):
    e = px[..., 3].astype(np.int32)
    f = np.where(e == 0, 0.0, np.ldexp(1.0, e - 136)).astype(np.float32)
    out = np.ones(px.shape, dtype=np.float32)
    out[..., :3] = px[..., :3].astype(np.float32) * f[..., None]
    return out


def write_hdr(path, px, rle):
    h, w = px.shape[:2]
    buf = io.BytesIO()
    buf.write(b"#?RADIANCE\n# written by tests/test_image_decode.py\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n")
    buf.write(f"-Y {h} +X {w}\n".encode())
    for y in range(h):
        if not rle:
            buf.write(px[y].tobytes())
            continue
        buf.write(bytes([2, 2, w >> 8, w & 255]))
        for c in range(4):
            row = px[y, :, c]
            i = 0
            while i < w:
                run = 1
                while i + run < w and run < 127 and row[i + run] == row[i]:
                    run += 1
                if run >= 3:
                    buf.write(bytes([128 + run, int(row[i])]))
                    i += run
                else:
                    j = i
                    while j < w and j - i < 128 and not (j + 2 < w and row[j] == row[j + 1] == row[j + 2]):
                        j += 1
                    j = max(j, i + 1)
                    buf.write(bytes([j - i]) + row[i:j].tobytes())
                    i = j
    path.write_bytes(buf.getvalue())


@pytest.mark.parametrize("size,rle", [((16, 9), True), ((16, 9), False), ((300, 7), True), ((5, 4), False), ((4, 3), True)])
def test_hdr_rgbe_exact(tmp_path, size, rle):
    w, h = size
    px = rgbe_pixels(w, h, w * 31 + h)
    f = tmp_path / "sky.hdr"
    write_hdr(f, px, rle and w >= 8)
    got = decode_hdr(f)
    assert got.shape == (h, w, 4)
    assert np.array_equal(got, rgbe_expected(px))


def test_hdr_errors(tmp_path):
    f = tmp_path / "bad.hdr"
    f.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_xyze\n\n-Y 2 +X 2\n" + bytes(16))
    with pytest.raises(RuntimeError, match="FORMAT"):
        decode_hdr(f)
    f.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n+Y 2 +X 2\n" + bytes(16))
    with pytest.raises(RuntimeError, match="orientation"):
        decode_hdr(f)
    f.write_bytes(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y 2 +X 2\n" + bytes(9))
    with pytest.raises(RuntimeError, match="truncated"):
        decode_hdr(f)
    f.write_bytes(b"P6 something else")
    with pytest.raises(RuntimeError, match="Radiance"):
        decode_hdr(f)


def test_scene_uses_a_real_hdr_sky_when_present(tmp_path):
    """SKY file.hdr with the file present goes through decode_hdr (linear, no sRGB step) and the
    texel the module sees is the RGBE value; with the file missing the synthetic probe is used."""
    from tracy_amd import Scene
    px = rgbe_pixels(32, 16, 9)
    (tmp_path / "data").mkdir()
    write_hdr(tmp_path / "data" / "probe.hdr", px, True)
    scn = tmp_path / "s.scn"
    scn.write_text("SCN hdrtest\nOUT 32 18\nCAM (0,0,5) (0,0,0) (0,1,0) 45\nSKY data/probe.hdr\nMTL m (0.5,0.5,0.5) 1 0\nOBJ S (0,0,0) 1 m\n")
    sc = Scene(str(scn), asset_root=str(tmp_path))
    assert sc.synthetic_textures == []
    d = sc.desc
    assert d.texture_count == 1 and d.textures[0].width == 32 and d.textures[0].height == 16
    tex = np.ctypeslib.as_array(d.textures[0].rgba, shape=(16, 32, 4))
    assert np.array_equal(tex, rgbe_expected(px))
    scn.write_text(scn.read_text().replace("probe.hdr", "absent.hdr"))
    sc2 = Scene(str(scn), asset_root=str(tmp_path))
    assert sc2.synthetic_textures == ["data/absent.hdr"]
    assert sc2.desc.textures[0].width == 1024
