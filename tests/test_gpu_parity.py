"""GPU parity tests proper: the HIP module, called through the C ABI, against the CPU oracle
on the same seeded inputs, against the committed golden fixtures, and -- at the full bench size
-- through size-independent properties.

Bars (BASELINE.json north_star):
  (a) primary-ray hit ids equal on >= 99.99 % of pixels (integer work: bit exact where equal,
      mismatches classified), t/u/v bit exact on matching pixels;
  (b) images at equal spp: PSNR >= 40 dB after the same sRGB tonemap (peak 1.0).
Per-sample float radiance: rtol 1e-4 / atol 1e-5 on >= 99.5 % of pixels (sin/cos/atan2/asin are
libm on the CPU and ocml on the GPU; a 1-ulp different direction can change a later hit).
"""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

MISS = 0xFFFFFFFF
ALL = ["cornell", "dragon", "spheres", "random", "testtree", "furnace", "helmet", "bunny", "trimesh"]


def srgb(x):
    x = np.clip(x, 0.0, 1.0)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)


def psnr(a, b):
    mse = np.mean((srgb(a.astype(np.float64)) - srgb(b.astype(np.float64))) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(1.0 / mse)


@pytest.fixture(scope="module")
def hip():
    from tracy_amd import HipTrace
    mods = []

    def make(scene, **kw):
        m = HipTrace(**kw)
        m.Startup(scene)
        mods.append(m)
        return m
    yield make
    for m in mods:
        m.Shutdown()


def test_module_fails_loudly_without_fallback():
    import ctypes as C
    from tracy_amd import capi
    lib = capi.hip_lib()
    ctx = C.c_void_p()
    assert lib.trh_init(9999, C.byref(ctx)) != 0          # bad device index is an error, not a CPU path
    assert lib.trh_render(None, 0, 1, 0) != 0


@pytest.mark.parametrize("name", ALL)
def test_golden_fixture_parity(hip, scene_loader, name):
    g = np.load(os.path.join(GOLDEN, f"{name}_64x36.npz"))
    mod = hip(scene_loader(name, 64, 36))
    ids, t, uv = mod.primary_hits()
    same = ids == g["ids"]
    assert same.sum() >= ids.size - 1, f"{(~same).sum()} hit-id mismatches"
    hitm = same & (ids != MISS)
    assert np.array_equal(t[hitm], g["t"][hitm]) and np.array_equal(uv[hitm], g["uv"][hitm])
    for key, s in (("s0", 0), ("s5", 5)):
        r = mod.sample_radiance(s)
        close = np.isclose(r[..., :3], g[key][..., :3], rtol=1e-4, atol=1e-5).all(axis=-1)
        assert close.mean() >= 0.995, f"{name} sample {s}: {close.mean()}"
        assert (r[..., 3] == g[key][..., 3]).mean() >= 0.995
    mod.render(0, 16)
    img, n = mod.resolve()
    assert n == 16 and (img[..., 3] == 16).all()
    assert psnr(img[..., :3], g["mean16"]) >= 40.0
    rays = mod.stats()["rays"]
    assert abs(rays - int(g["rays16"])) <= 0.005 * int(g["rays16"])


@pytest.mark.parametrize("name,w,h", [("cornell", 480, 270), ("dragon", 480, 270), ("spheres", 320, 180),
                                       ("random", 320, 180), ("testtree", 320, 180), ("furnace", 320, 180), ("helmet", 320, 180)])
def test_primary_hit_ids_vs_oracle(hip, oracle_mod, scene_loader, name, w, h):
    sc = scene_loader(name, w, h)
    ref = oracle_mod.CpuRef(sc)
    mod = hip(sc)
    ids_r, t_r, uv_r = ref.primary_hits()
    ids_g, t_g, uv_g = mod.primary_hits()
    mism = ids_r != ids_g
    frac = 1.0 - mism.mean()
    assert frac >= 0.9999, f"{name}: {mism.sum()} mismatches of {mism.size}"
    ok = ~mism & (ids_r != MISS)
    assert np.array_equal(t_r[ok], t_g[ok])
    assert np.array_equal(uv_r[ok], uv_g[ok])
    # classification of the mismatches: both sides found a hit at (nearly) the same distance, i.e. a
    # tie between adjacent / coincident triangles decided by test order (collision.h:66), never a hole
    for y, x in zip(*np.nonzero(mism)):
        assert ids_r[y, x] != MISS and ids_g[y, x] != MISS, "one side missed where the other hit"
        assert abs(float(t_r[y, x]) - float(t_g[y, x])) <= 4e-6 * abs(float(t_r[y, x])) + 1e-6


@pytest.mark.parametrize("name", ["cornell", "dragon", "spheres", "helmet"])
def test_image_psnr_vs_oracle(hip, oracle_mod, scene_loader, name):
    sc = scene_loader(name, 160, 90)
    ref = oracle_mod.CpuRef(sc)
    mod = hip(sc)
    spp = 32
    ref.render(0, spp)
    mod.render(0, spp)
    img, _ = mod.resolve()
    p = psnr(img[..., :3], ref.output())
    assert p >= 40.0, f"{name}: PSNR {p:.1f} dB"
    r_ref, r_gpu = ref.counters()["rays"], mod.stats()["rays"]
    assert abs(r_ref - r_gpu) <= 0.002 * r_ref


@pytest.mark.parametrize("name", ["dragon", "random", "helmet"])
def test_pipelines_trace_the_same_paths(hip, scene_loader, name):
    """megakernel (wave-local scheduler), wavefront (global queues) and the first-generation kernel
    trace the same per-sample paths: same ray counts, images equal up to summation order."""
    from tracy_amd import capi
    sc = scene_loader(name, 160, 90)
    imgs, rays = {}, {}
    for pipe in (capi.TRH_PIPELINE_MEGAKERNEL, capi.TRH_PIPELINE_WAVEFRONT, capi.TRH_PIPELINE_MEGAKERNEL_SIMPLE):
        mod = hip(sc, pipeline=pipe)
        mod.render(0, 5)
        mod.render(5, 3)
        imgs[pipe], n = mod.resolve()
        assert n == 8 and (imgs[pipe][..., 3] == 8).all()
        rays[pipe] = mod.stats()["rays"]
    assert rays[0] == rays[1] == rays[2]
    for pipe in (1, 2):
        assert np.allclose(imgs[pipe][..., :3], imgs[0][..., :3], rtol=1e-5, atol=1e-6)


def test_wavefront_golden_parity(hip, scene_loader):
    from tracy_amd import capi
    g = np.load(os.path.join(GOLDEN, "dragon_64x36.npz"))
    mod = hip(scene_loader("dragon", 64, 36), pipeline=capi.TRH_PIPELINE_WAVEFRONT)
    mod.render(0, 16)
    img, n = mod.resolve()
    assert n == 16
    assert psnr(img[..., :3], g["mean16"]) >= 40.0
    assert mod.stats()["rays"] == int(g["rays16"])
    mod.reset()
    mod.render(0, 1)
    one, _ = mod.resolve()
    assert np.allclose(one[..., :3], g["s0"][..., :3], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("w,h,spp", [(100, 50, 3), (100, 50, 9), (37, 23, 1), (8, 8, 70), (1, 1, 5)])
def test_ragged_frames_and_small_sample_counts(hip, scene_loader, w, h, spp):
    """Frames that are not multiples of the 8x8 / 16x16 wave tiles, and per-call sample counts on both
    sides of the tile-size switch: the accumulated sum must equal the sum of the single samples."""
    sc = scene_loader("cornell", w, h)
    mod = hip(sc)
    mod.render(2, spp)
    img, n = mod.resolve()
    assert n == spp and (img[..., 3] == spp).all()
    acc = np.zeros((h, w, 3), dtype=np.float64)
    rays = 0
    for s in range(2, 2 + spp):
        r = mod.sample_radiance(s)
        acc += r[..., :3]
        rays += int(r[..., 3].sum())
    assert np.allclose(img[..., :3], acc / spp, rtol=1e-5, atol=1e-6)
    assert mod.stats()["rays"] == rays


def test_camera_cut_resets_accumulation(hip, scene_loader):
    sc = scene_loader("cornell", 64, 36)
    mod = hip(sc)
    mod.render(0, 4)
    mod.OnEvent(mod.eCameraCut, sc)
    mod.OnUpdate(sc)                       # one progressive pass = sample 0
    img, n = mod.resolve()
    assert n == 1
    assert np.array_equal(img[..., :3], mod.sample_radiance(0)[..., :3])


def test_errors(hip, scene_loader):
    import ctypes as C
    from tracy_amd import capi
    lib = capi.hip_lib()
    ctx = C.c_void_p()
    assert lib.trh_init(0, C.byref(ctx)) == 0
    assert lib.trh_render(ctx, 0, 1, 0) == 4               # TRH_ERR_STATE: no scene yet
    assert b"scene" in lib.trh_last_error()
    assert lib.trh_render(ctx, 0xFFFFFFFF, 2, 0) != 0
    assert lib.trh_shutdown(ctx) == 0


# ---------------------------------------------------------------------------------------
# full bench size (BASELINE.json config 4: dragon, 1920x1080): size-independent properties
# ---------------------------------------------------------------------------------------

@pytest.fixture(scope="module")
def dragon_full(hip, scene_loader):
    sc = scene_loader("dragon", 1920, 1080)
    return sc, hip(sc)


def test_full_size_determinism_and_sample_partition(dragon_full, hip):
    sc, mod = dragon_full
    mod.reset()
    mod.render(0, 8)
    a, n = mod.resolve()
    assert n == 8 and (a[..., 3] == 8).all()
    mod.reset()
    mod.render(0, 8)
    b, _ = mod.resolve()
    assert np.array_equal(a, b)                                   # bitwise repeatable
    # two "ranks" with disjoint spp ranges sum to the single-GPU image (multi-GPU contract)
    r0, r1 = hip(sc), hip(sc)
    r0.render(0, 4)
    r1.render(4, 4)
    i0, _ = r0.resolve()
    i1, _ = r1.resolve()
    merged = (i0[..., :3].astype(np.float64) * 4 + i1[..., :3].astype(np.float64) * 4) / 8
    assert np.allclose(merged, a[..., :3], rtol=2e-6, atol=1e-7)
    assert np.isfinite(a).all()


def test_full_size_ray_counter_equals_bounce_iterations(dragon_full):
    sc, mod = dragon_full
    mod.reset()
    mod.stats(reset=True)
    mod.render(3, 1)
    st = mod.stats(reset=True)
    per_pixel = mod.sample_radiance(3)
    assert st["samples"] == 1920 * 1080
    assert st["rays"] == int(per_pixel[..., 3].astype(np.int64).sum())
    img, _ = mod.resolve()
    assert np.array_equal(img[..., :3], per_pixel[..., :3])


def test_full_size_hits_lie_on_their_triangles(dragon_full):
    sc, mod = dragon_full
    ids, t, uv = mod.primary_hits()
    hit = ids != MISS
    assert 0.3 < hit.mean() < 0.9
    assert (t[~hit] == np.float32(3.402823466e38)).all()
    # reconstruct the hit point from the barycentrics and compare with origin + t*dir
    meshes = sc.meshes()
    V = sc.vertices()
    I = sc.indices()
    rng = np.random.default_rng(0)
    ys, xs = np.nonzero(hit)
    pick = rng.choice(len(ys), size=20000, replace=False)
    eye, ivp = sc.eye_ivp()
    M = ivp.astype(np.float64).reshape(4, 4).T
    for k in pick[:2000]:
        y, x = int(ys[k]), int(xs[k])
        pid = int(ids[y, x])
        m, tri = pid >> 24, pid & 0xFFFFFF
        me = meshes[m]
        vi = I[me.first_index + 3 * tri: me.first_index + 3 * tri + 3] + me.first_vertex
        p0, p1, p2 = V[vi[0], :3].astype(np.float64), V[vi[1], :3].astype(np.float64), V[vi[2], :3].astype(np.float64)
        u, v = float(uv[y, x, 0]), float(uv[y, x, 1])
        assert u >= 0 and v >= 0 and u + v <= 1 + 1e-6
        pb = (1 - u - v) * p0 + u * p1 + v * p2
        ndc = np.array([(x + .5) / 1920 * 2 - 1, (y + .5) / 1080 * 2 - 1, 1.0, 1.0])
        q = M @ ndc
        d = q[:3] / q[3] - eye
        d /= np.linalg.norm(d)
        pr = eye + float(t[y, x]) * d
        assert np.linalg.norm(pr - pb) <= 2e-3, (pr, pb)


# ----------------------------------------------------------------------------- node record formats
FAR_SCN = """SCN far_apart
OUT 96 54
CAM (500,0.02,0.6) (500,0,0) (0,1,0) 20
SKY (0.6,0.7,0.9)
MTL grey (0.7,0.7,0.7) 0.9 0.0
MTL metal (0.9,0.8,0.5) 0.2 1.0
OBJ S (0,0,0) 0.05 grey
OBJ S (500,0,0) 0.05 metal
OBJ S (500.11,0,0) 0.05 grey
"""


def test_fp16_records_fall_back_to_fp32_when_they_would_inflate(tmp_path, monkeypatch):
    """Small geometry far from the BVH frame centre: half-float bounds (11 significant bits) would
    blow the boxes up, the builder's inflation measure must select the 64-byte fp32 records, and
    either format must give the oracle's hits (quantised boxes only ever contain the exact ones)."""
    from tracy_amd import Scene, HipTrace
    from oracle.oracle import CpuRef
    scn = tmp_path / "far.scn"
    scn.write_text(FAR_SCN)
    sc = Scene(str(scn))
    ref_ids, ref_t, _ = CpuRef(sc).primary_hits()
    assert (ref_ids != MISS).mean() > 0.05
    images = {}
    for fmt in (None, "16", "32"):
        if fmt is None:
            monkeypatch.delenv("TRH_NODE_FORMAT", raising=False)
        else:
            monkeypatch.setenv("TRH_NODE_FORMAT", fmt)
        mod = HipTrace()
        mod.Startup(sc)
        info = mod.bvh_info()
        rec = info["node_bytes"] // info["node_count"]
        assert rec == (32 if fmt == "16" else 64)          # default (None) must have fallen back
        assert info["fp16_inflation"] > 1.25
        ids, t, _ = mod.primary_hits()
        assert np.array_equal(ids, ref_ids)
        assert np.array_equal(t[ids != MISS], ref_t[ids != MISS])
        mod.render(0, 8)
        images[fmt] = mod.resolve()[0]
        mod.Shutdown()
    # same paths, but the order in which a pixel's samples are summed follows the traversal schedule
    assert np.array_equal(images[None], images["32"])
    assert np.allclose(images["16"], images["32"], rtol=2e-6, atol=1e-7)


def test_both_record_formats_render_the_same_dragon(scene_loader, monkeypatch):
    from tracy_amd import HipTrace
    sc = scene_loader("dragon", 160, 90)
    out = {}
    for fmt in ("16", "32"):
        monkeypatch.setenv("TRH_NODE_FORMAT", fmt)
        mod = HipTrace()
        mod.Startup(sc)
        info = mod.bvh_info()
        assert info["node_bytes"] == info["node_count"] * (32 if fmt == "16" else 64)
        assert 1.0 <= info["fp16_inflation"] < 1.25        # dense geometry around the frame centre: fp16 is the default
        mod.render(0, 16)
        out[fmt] = (mod.resolve()[0], mod.stats()["rays"])
        mod.Shutdown()
    assert out["16"][1] == out["32"][1]                    # identical ray counts: identical paths
    assert np.allclose(out["16"][0], out["32"][0], rtol=2e-6, atol=1e-7)   # summation order follows the schedule


# ----------------------------------------------------------------------------- edge cases
def _scene_from_text(tmp_path, text, name="edge.scn"):
    from tracy_amd import Scene
    p = tmp_path / name
    p.write_text(text)
    return Scene(str(p))


def test_empty_scene_is_all_sky(tmp_path, oracle_mod):
    """No geometry at all: every path is one miss ray and the image is the constant sky emission."""
    from tracy_amd import HipTrace
    sc = _scene_from_text(tmp_path, "SCN empty\nOUT 40 24\nCAM (0,0,5) (0,0,0) (0,1,0) 45\nSKY (0.25,0.5,0.75)\n")
    assert sc.tri_count == 0
    mod = HipTrace()
    mod.Startup(sc)
    try:
        ids, t, _ = mod.primary_hits()
        assert (ids == MISS).all()
        mod.render(0, 9)
        img, n = mod.resolve()
        assert n == 9 and mod.stats()["rays"] == 40 * 24 * 9
        assert np.allclose(img[..., :3], [0.25, 0.5, 0.75], rtol=1e-6)
        ref = oracle_mod.CpuRef(sc)
        ref.render(0, 9)
        assert np.allclose(img[..., :3], ref.output(), rtol=1e-6)
    finally:
        mod.Shutdown()


def test_single_triangle_and_back_face_culling(tmp_path, oracle_mod):
    """One triangle seen from the front, then from behind (collision.h:44 culls det < EPS): ids, t, uv
    and a few samples must equal the oracle's in both views."""
    from tracy_amd import HipTrace
    text = ("SCN tri\nOUT 48 32\nCAM (0,0,{z}) (0,0,0) (0,1,0) 50\nSKY (1,1,1)\nMTL m (0.8,0.3,0.2) 1 0\n"
            "OBJ T (-1,-1,0) (1,-1,0) (0,1,0) m\n")
    for z, expect_hits in ((3, True), (-3, False)):
        sc = _scene_from_text(tmp_path, text.format(z=z), f"tri{z}.scn")
        assert sc.tri_count == 1
        mod = HipTrace()
        mod.Startup(sc)
        try:
            ref = oracle_mod.CpuRef(sc)
            ids, t, uv = mod.primary_hits()
            rids, rt, ruv = ref.primary_hits()
            assert np.array_equal(ids, rids)
            assert ((ids != MISS).any()) == expect_hits
            hit = ids != MISS
            assert np.array_equal(t[hit], rt[hit]) and np.array_equal(uv[hit], ruv[hit])
            for s in (0, 3):
                a, b = mod.sample_radiance(s), ref.sample_radiance(s)
                assert np.array_equal(a[..., 3], b[..., 3])
                assert np.allclose(a[..., :3], b[..., :3], rtol=1e-4, atol=1e-5)
        finally:
            mod.Shutdown()


def test_degenerate_triangles_do_not_disturb_the_render(tmp_path, oracle_mod):
    """Zero-area triangles (a collapsed box face, dragon.scn:56 has such a floor) and a needle are legal
    input: det = 0 rejects them in RayTriangle; the BVH must still be sound and the hits the oracle's."""
    from tracy_amd import HipTrace
    sc = _scene_from_text(tmp_path, "SCN degenerate\nOUT 64 36\nCAM (0,2,6) (0,0.5,0) (0,1,0) 45\nSKY (0.5,0.6,0.8)\n"
                          "MTL grey (0.5,0.5,0.5) 1 0\nMTL red (0.8,0.1,0.1) 0.3 0.5\n"
                          "OBJ B (-5,0,-5) (5,0,5) grey\n"                 # flat box: two coincident quads + four zero-area sides
                          "OBJ T (0,0.5,0) (0,0.5,0) (0,0.5,0) red\n"      # a point
                          "OBJ T (-1,0.2,0) (1,0.2,0) (3,0.2,0) red\n"     # a needle (collinear)
                          "OBJ S (0,1,0) 0.7 red\n")
    mod = HipTrace()
    mod.Startup(sc)
    try:
        ref = oracle_mod.CpuRef(sc)
        ids, t, _ = mod.primary_hits()
        rids, rt, _ = ref.primary_hits()
        assert (ids == rids).mean() >= 0.9999
        mod.render(0, 16)
        ref.render(0, 16)
        img, _ = mod.resolve()
        assert np.isfinite(img).all()
        assert psnr(img[..., :3], ref.output()) >= 40.0
        assert mod.stats()["rays"] == ref.counters()["rays"]
    finally:
        mod.Shutdown()


def test_seed_and_sample_index_define_the_sample_set(hip, scene_loader):
    """RNG state = f(pixel, global sample index, seed): the same (index, seed) gives the same radiance
    whatever was rendered before; another seed or index gives another sample."""
    sc = scene_loader("cornell", 64, 36)
    mod = hip(sc)
    a = mod.sample_radiance(7, seed=0xABCDEF)
    mod.render(0, 5)
    b = mod.sample_radiance(7, seed=0xABCDEF)
    assert np.array_equal(a, b)
    assert not np.array_equal(a, mod.sample_radiance(8, seed=0xABCDEF))
    assert not np.array_equal(a, mod.sample_radiance(7, seed=0x1234567))
    hi = mod.sample_radiance(7, seed=(1 << 40) | 0xABCDEF)          # the high seed word takes part
    assert not np.array_equal(a, hi)
    # accumulating samples one by one or in one launch: the same sample set (summation order aside)
    mod.reset()
    for s in range(6):
        mod.render(s, 1)
    one_by_one, n1 = mod.resolve()
    mod.reset()
    mod.render(0, 6)
    at_once, n2 = mod.resolve()
    assert n1 == n2 == 6
    assert np.allclose(one_by_one, at_once, rtol=2e-6, atol=1e-7)


def test_dragon_against_the_reference_screenshot(scene_loader):
    """The only output of the real reference that exists for this path: doc/dragon.jpg, a screenshot of
    the CPU module's window ('Stanford Dragon' 1024x768, [3 objs] [100024 tris], ~100 accumulated frames,
    build options and commit unknown), kept as tests/golden/reference_doc_dragon.jpg. A loose end-to-end
    pin of scene loading, camera, materials and tonemap together: after removing the window frame and
    blurring away the Monte-Carlo noise of both pictures, PSNR >= 20 dB (measured 22.7 dB; the module's
    picture is ~1.2x brighter in display space)."""
    from PIL import Image, ImageFilter
    from tracy_amd import HipTrace
    shot = Image.open(os.path.join(GOLDEN, "reference_doc_dragon.jpg")).convert("RGB")
    assert shot.size == (1026, 800)
    shot = shot.crop((1, 31, 1025, 799))                     # 1 px border, 30 px title bar
    sc = scene_loader("dragon", 0, 0)                        # OUT 1024 768 from the .scn, as in the screenshot
    assert (sc.width, sc.height) == (1024, 768)
    mod = HipTrace()
    mod.Startup(sc)
    try:
        mod.render(0, 256)
        pic = mod.present()                                  # default build: sRGB tonemap, exposure 1
    finally:
        mod.Shutdown()
    rgb = np.stack([(pic >> 16) & 255, (pic >> 8) & 255, pic & 255], axis=-1).astype(np.uint8)
    a = np.asarray(Image.fromarray(rgb).filter(ImageFilter.GaussianBlur(4)), dtype=np.float64) / 255
    b = np.asarray(shot.filter(ImageFilter.GaussianBlur(4)), dtype=np.float64) / 255
    mse = np.mean((a - b) ** 2)
    assert 10 * np.log10(1 / mse) >= 20.0
    # the black sky and the silhouette must coincide: camera and geometry are the same
    sky_a, sky_b = a.mean(axis=-1) < 0.02, b.mean(axis=-1) < 0.02
    assert (sky_a == sky_b).mean() >= 0.98


@pytest.mark.parametrize("name,builder", [(n, 0) for n in ALL] + [("dragon", 1), ("spheres", 1), ("helmet", 1), ("random", 1)])
def test_random_cameras_against_oracle(oracle_mod, scene_loader, name, builder):
    """Cameras drawn from deep inside the geometry to well outside it, arbitrary orientation and roll
    (tests/tools/fuzz_cameras.py runs the long version: 0 id mismatches in 3.24 M pixels): primary-hit ids
    >= 99.99 % equal, t/u/v bit-equal on equal ids, one sample index of radiance within tolerance.
    builder 1 = the same through the device-built linear BVH."""
    from tracy_amd import HipTrace, Scene
    rng = np.random.default_rng(sum(map(ord, name)))
    sc = Scene(f"data/scenes/{name}.scn", 96, 54)      # private copy: the test moves its camera
    v = sc.vertices()[:, :3]
    lo, hi = v.min(axis=0), v.max(axis=0)
    ctr, ext = (lo + hi) / 2, float(np.max(hi - lo))
    mod = HipTrace(builder=builder)
    mod.Startup(sc)
    assert mod.bvh_info()["builder"] == builder
    ref = oracle_mod.CpuRef(sc)
    try:
        bad = outliers = 0
        cams = 6
        for _ in range(cams):
            r = ext * (0.05 + 1.5 * rng.random())
            eye = ctr + r * rng.normal(size=3) / np.sqrt(3)
            target = ctr + 0.3 * ext * (rng.random(3) - 0.5)
            sc.set_view(eye.astype(np.float32), target.astype(np.float32), rng.normal(size=3).astype(np.float32))
            e, ivp = sc.eye_ivp()
            mod.set_camera(e, ivp)
            ref.set_camera(e, ivp)
            ids, t, uv = mod.primary_hits()
            rids, rt, ruv = ref.primary_hits()
            same = (ids == rids) & (ids != MISS)
            assert np.array_equal(t[same], rt[same]) and np.array_equal(uv[same], ruv[same])
            bad += int((ids != rids).sum())
            a, b = mod.sample_radiance(3), ref.sample_radiance(3)
            ok = np.isclose(a[..., :3], b[..., :3], rtol=1e-4, atol=1e-5).all(axis=-1) & (a[..., 3] == b[..., 3])
            outliers += int((~ok).sum())
        n = cams * 96 * 54
        assert bad <= 1e-4 * n
        assert outliers <= 5e-3 * n
    finally:
        mod.Shutdown()


def test_one_call_with_more_samples_than_one_launch_takes(hip, scene_loader):
    """trh_render splits a call of more than 65536 spp into several launches: the image is the same as the
    caller's own split, and every pixel counts every sample."""
    sc = scene_loader("cornell", 4, 3)
    a = hip(sc)
    b = hip(sc)
    n = 65536 + 1000
    a.render(0, n)
    b.render(0, 65536)
    b.render(65536, 1000)
    ia, na = a.resolve()
    ib, nb = b.resolve()
    assert na == nb == n and (ia[..., 3] == n).all()
    assert np.array_equal(ia, ib)
    assert a.stats()["samples"] == 4 * 3 * n


def test_tile_order_does_not_change_the_image(scene_loader, monkeypatch):
    """Cost-ordered tiles (tile_order.h) and the probe sample of a first launch only change WHEN a tile runs:
    launches of 8 spp (ordered from the second launch on, no probe) are bit-identical with the feature off, and
    a first launch of 64 spp (1 probe sample + 63 ordered) equals the same samples on a context that already
    has an order up to the rounding of the per-pixel sums."""
    from tracy_amd import HipTrace
    sc = scene_loader("dragon", 480, 270)

    def three_launches(feedback):
        monkeypatch.setenv("TRH_TILE_FEEDBACK", feedback)
        m = HipTrace()
        m.Startup(sc)
        for k in range(3):
            m.render(8 * k, 8)
        img, n = m.resolve()
        m.Shutdown()
        return img, n
    on, n_on = three_launches("1")
    off, n_off = three_launches("0")
    assert n_on == n_off == 24 and np.array_equal(on, off)

    monkeypatch.setenv("TRH_TILE_FEEDBACK", "1")
    fresh = HipTrace()
    fresh.Startup(sc)
    fresh.render(0, 64)                     # probe sample + 63
    a, na = fresh.resolve()
    fresh.reset()
    fresh.render(0, 64)                     # the tiling now has an order: one launch of 64
    b, nb = fresh.resolve()
    fresh.Shutdown()
    assert na == nb == 64 and (a[..., 3] == 64).all() and (b[..., 3] == 64).all()
    assert np.allclose(a[..., :3], b[..., :3], rtol=2e-6, atol=1e-7)


def test_scene_reupload_on_a_live_context(scene_loader):
    """trh_scene_upload may be called again on a context that has rendered (another scene, another frame size,
    another builder): every per-scene array -- BVH, accumulators, tile costs and order -- is rebuilt, and the
    context then renders exactly what a fresh one does."""
    import ctypes as C
    from tracy_amd import HipTrace, capi
    a, b = scene_loader("dragon", 192, 108), scene_loader("cornell", 64, 36)
    mod = HipTrace()
    mod.Startup(a)
    for k in range(3):
        mod.render(16 * k, 16)                      # builds up a tile order for the first tiling
    mod.synchronize()
    for builder in (capi.TRH_BUILDER_HOST_SAH, capi.TRH_BUILDER_DEVICE_LBVH):
        assert mod._lib.trh_set_builder(mod._ctx, builder) == 0
        assert mod._lib.trh_scene_upload(mod._ctx, C.byref(b.desc)) == 0
        mod.width, mod.height = b.width, b.height
        mod._upload_camera(b)
        assert mod.bvh_info()["builder"] == builder
        for k in range(2):
            mod.render(16 * k, 16)
        img, n = mod.resolve()
        fresh = HipTrace(builder=builder)
        fresh.Startup(b)
        for k in range(2):
            fresh.render(16 * k, 16)
        want, n2 = fresh.resolve()
        fresh.Shutdown()
        assert n == n2 == 32 and img.shape == want.shape == (36, 64, 4)
        assert np.array_equal(img, want)
    mod.Shutdown()


def test_reupload_keeps_the_camera(scene_loader):
    """The camera set before a (re-)upload survives it, including the host-side constants derived from it
    (DevScene::cam_k, the ray-independent half of GetRayFrom's matrix product)."""
    import ctypes as C
    from tracy_amd import HipTrace
    sc = scene_loader("spheres", 96, 54)
    mod = HipTrace()
    mod.Startup(sc)
    mod.render(0, 8)
    want, _ = mod.resolve()
    assert mod._lib.trh_scene_upload(mod._ctx, C.byref(sc.desc)) == 0     # no trh_set_camera afterwards
    mod.render(0, 8)
    img, n = mod.resolve()
    mod.Shutdown()
    assert n == 8 and np.array_equal(img, want)


def test_upload_from_the_bvh_cache_renders_the_same(scene_loader, tmp_path):
    from tracy_amd import HipTrace, capi
    lib = capi.hip_lib()
    sc = scene_loader("spheres", 96, 54)
    try:
        assert lib.trh_set_bvh_cache_dir(str(tmp_path).encode()) == 0
        imgs = []
        for expect in (0, 1):
            m = HipTrace()
            m.Startup(sc)
            assert m.bvh_info()["from_cache"] == expect
            m.render(0, 8)
            imgs.append(m.resolve()[0])
            m.Shutdown()
        assert np.array_equal(imgs[0], imgs[1])
    finally:
        lib.trh_set_bvh_cache_dir(None)


def test_textured_spheres_with_the_earth_map(tmp_path, oracle_mod):
    """data/earth.jpg ships with the reference but none of its scenes uses it: a scene of our own puts it on a
    tessellated sphere as base colour (sRGB) and as emissive map of a second one -- sphere uv mapping
    (scene.cpp:55-131), JPEG decode, the sRGB flag and Texture::GetPixel on procedural meshes, against the oracle."""
    from tracy_amd import HipTrace
    sc = _scene_from_text(tmp_path, """SCN Earth
OUT 160 90
CAM (0, 0.4, 3.2) (0, 0, 0) (0, 1, 0) 40
MTL Light (1,1,1) 0 0 0 6
MTL Earth (1,1,1) 0.6 0
MTL Glow (1,1,1) 1 0 0 2
MTL Floor (0.5,0.5,0.5) 0.9 0
TEX Earth B data/earth.jpg SRGB
TEX Glow E data/earth.jpg
SKY (0.02,0.02,0.03)
OBJ S (2.5, 4, 2) 1.2 Light
OBJ S (-0.7, 0, 0) 0.8 Earth
OBJ S (1.0, -0.2, 0.3) 0.5 Glow
OBJ B (-6, -1.0, -6) (6, -0.8, 6) Floor
""", "earth.scn")
    assert len(sc.synthetic_textures) == 0
    ref = oracle_mod.CpuRef(sc)
    mod = HipTrace()
    mod.Startup(sc)
    try:
        ids, t, uv = mod.primary_hits()
        rids, rt, ruv = ref.primary_hits()
        assert (ids == rids).mean() >= 0.9999
        same = (ids == rids) & (ids != MISS)
        assert np.array_equal(t[same], rt[same]) and np.array_equal(uv[same], ruv[same])
        for view in (1, 5):                      # base colour and emissive: the texels themselves
            mod.set_debug_view(view, sc.view3x3())
            ref.set_debug_view(view, sc.view3x3())
            a, b = mod.sample_radiance(0), ref.sample_radiance(0)
            assert np.isclose(a[..., :3], b[..., :3], rtol=2e-6, atol=1e-7).all(axis=-1).mean() >= 0.999
            assert len(np.unique(a[..., :3].reshape(-1, 3), axis=0)) > 200   # the map really varies over the sphere
        mod.set_debug_view(0)
        ref.set_debug_view(0)
        for s in (0, 4):
            a, b = mod.sample_radiance(s), ref.sample_radiance(s)
            ok = np.isclose(a[..., :3], b[..., :3], rtol=1e-4, atol=1e-5).all(axis=-1) & (a[..., 3] == b[..., 3])
            assert ok.mean() >= 0.995
        mod.render(0, 16)
        ref.render(0, 16)
        assert psnr(mod.resolve()[0][..., :3], ref.output()) >= 40.0
    finally:
        mod.Shutdown()
