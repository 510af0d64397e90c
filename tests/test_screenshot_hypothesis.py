"""The one output of the real reference for this path is doc/dragon.jpg (tests/golden/reference_doc_dragon.jpg), a
screenshot of the CPU module's window. Today's Scatter gives a picture ~1.2x brighter in display space than that
screenshot. Hypothesis: the screenshot predates the 10 % white specular lobe of non-metals
(`materialspecularchance = lerp(.1f, 1.f, metalness)`, src/material.h:252-260). Test: the ORACLE with that floor set
to 0 (test-only switch oref_set_matchance0) must land on the screenshot -- and it does: at 64 spp the blurred PSNR goes
from 25.6 to 35.6 dB and the whole-picture display mean from 0.232 to 0.2005 against the screenshot's 0.1999.
Everything else -- scene loading, camera, geometry, the other BSDF terms, accumulation, tonemap -- is therefore
confirmed end to end by a picture the reference itself made.

The comparison depends on the sample count: the scene is lit by a small emitter, per-pixel radiance is heavy-tailed and
the sRGB curve is concave, so the DISPLAY-space mean keeps rising with the number of accumulated frames (floor region
0.354 / 0.419 / 0.434 / 0.451 / 0.464 at 24 / 48 / 64 / 100 / 256 spp without the lobe; screenshot 0.421). The
screenshot's frame count is not known exactly (its title bar shows ~1.2 fps), so the test uses 64
spp and bounds, not equalities: with the lobe the picture is too bright at EVERY sample count >= 48 (PSNR 27.1 / 25.6 /
24.2 / 23.2 dB), without it it matches to 33.9 / 35.6 / 33.5 / 30.7 dB."""
import os

import numpy as np

from conftest import GOLDEN


def test_dragon_screenshot_matches_once_the_white_specular_floor_is_removed(oracle_mod, scene_loader):
    from PIL import Image, ImageFilter
    shot = Image.open(os.path.join(GOLDEN, "reference_doc_dragon.jpg")).convert("RGB")
    assert shot.size == (1026, 800)
    shot = shot.crop((1, 31, 1025, 799))                    # 1 px border, 30 px title bar
    want = np.asarray(shot.filter(ImageFilter.GaussianBlur(4)), dtype=np.float64) / 255
    sc = scene_loader("dragon", 0, 0)                       # OUT 1024 768 from the .scn, as in the screenshot
    L = oracle_mod.lib()
    res = {}
    try:
        for floor0 in (0, 1):
            L.oref_set_matchance0(floor0)
            ref = oracle_mod.CpuRef(sc)
            ref.render(0, 64)
            pic = oracle_mod.tonemap_bgra8(ref.output(), mode=1)       # default build: sRGB, exposure 1
            ref.close()
            rgb = np.stack([(pic >> 16) & 255, (pic >> 8) & 255, pic & 255], axis=-1).astype(np.uint8)
            got = np.asarray(Image.fromarray(rgb).filter(ImageFilter.GaussianBlur(4)), dtype=np.float64) / 255
            res[floor0] = (10 * np.log10(1 / np.mean((got - want) ** 2)), got[700:740, 100:300].mean() / want[700:740, 100:300].mean(),
                           ((got.mean(axis=-1) < 0.02) == (want.mean(axis=-1) < 0.02)).mean())
    finally:
        L.oref_set_matchance0(0)
    (psnr_ref, floor_ref, sky_ref), (psnr_0, floor_0, sky_0) = res[0], res[1]
    assert sky_ref >= 0.98 and sky_0 >= 0.98                # silhouette and black sky coincide either way: camera + geometry
    assert 23.0 <= psnr_ref <= 28.0                         # today's BRDF: brighter than the screenshot (measured 25.6 dB, floor x1.21)
    assert 1.12 <= floor_ref <= 1.30
    assert psnr_0 >= 33.0 and psnr_0 >= psnr_ref + 7.0      # without the floor: measured 35.6 dB
    assert abs(floor_0 - 1.0) <= 0.06                       # measured 1.03


def test_cornell_screenshot_prefers_todays_scatter(oracle_mod, scene_loader):
    """Round 3: a second picture the reference made, doc/cornell.jpg (tests/golden/reference_doc_cornell.jpg: "Tracy 2.0 (CPU)
    'Cornell' 800x800, elapsed 9 min at 5.4 fps", i.e. converged). Its scene had two spheres today's cornell.scn no longer
    has ("10 objs, 4.06K tris" against 8 boxes / 96 triangles), but walls, light, blocks and camera coincide, so the wall,
    ceiling and floor regions away from the spheres can be compared. Result: they match TODAY's Scatter -- 10 % white
    specular lobe included -- to 1-4 % in display space, and with the lobe removed the coloured walls come out 14-20 % too
    dark (their off-colour channels are almost entirely that lobe: red wall G 0.109 against the screenshot's 0.184).
    So this picture was made with `materialspecularchance = lerp(.1f, 1.f, metalness)` (material.h:252-260) in place and
    pins the oracle's BRDF, accumulation and tonemap end to end, loosely. It also says that round 2's reading of
    doc/dragon.jpg ("predates the lobe") is not the only one: a floor material of lower albedo in the author's dragon.scn of
    that day explains the dragon picture equally (the lobe and the albedo are indistinguishable on a roughness-1 dielectric,
    and the metal dragon itself matches either way)."""
    from PIL import Image, ImageFilter
    from scipy.ndimage import gaussian_filter
    shot = Image.open(os.path.join(GOLDEN, "reference_doc_cornell.jpg")).convert("RGB")
    assert shot.size == (802, 832)
    shot = shot.crop((1, 31, 801, 831))                     # 1 px border, 30 px title bar
    want = np.asarray(shot.filter(ImageFilter.GaussianBlur(4)), dtype=np.float64) / 255
    sc = scene_loader("cornell", 0, 0)                      # OUT 800 800 from the .scn

    def srgb(x):
        x = np.clip(x, 0, 1)
        return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)

    regions = {"green wall": (slice(300, 500), slice(40, 120)), "red wall": (slice(300, 500), slice(690, 760)),
               "back wall top": (slice(215, 280), slice(220, 580)), "ceiling": (slice(60, 120), slice(480, 700)),
               "floor front left": (slice(720, 770), slice(80, 300))}
    L = oracle_mod.lib()
    ratio = {}
    try:
        for floor0 in (0, 1):
            L.oref_set_matchance0(floor0)
            ref = oracle_mod.CpuRef(sc)
            ref.render(0, 96)
            lin = ref.output()[::-1]                        # row 0 is the bottom of the picture in the module
            ref.close()
            # blurred in LINEAR space before the display curve: 96 spp against the screenshot's ~2900 frames
            got = srgb(np.stack([gaussian_filter(lin[..., c].astype(np.float64), 4) for c in range(3)], axis=-1))
            ratio[floor0] = {k: got[ys, xs].mean(axis=(0, 1)) / want[ys, xs].mean(axis=(0, 1)) for k, (ys, xs) in regions.items()}
    finally:
        L.oref_set_matchance0(0)
    for k in regions:                                       # today's Scatter: every region, every channel within 8 % (measured <= 7 %)
        assert np.all(np.abs(ratio[0][k] - 1.0) <= 0.08), (k, ratio[0][k])
    # without the lobe the off-colour channels of the coloured walls collapse (measured 0.59 / 0.82 and 0.59 / 0.58)
    assert ratio[1]["red wall"][1] <= 0.70 and ratio[1]["red wall"][2] <= 0.70
    assert ratio[1]["green wall"][0] <= 0.88 and ratio[1]["green wall"][2] <= 0.90
