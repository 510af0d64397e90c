"""Display-side post: CPUDetails::Tonemap + Bitmap::SetPixel.

CPU part: the oracle's restatement against hand-computed values of the formulas in
cpu_details.cpp:218-243 / cuda_details.cpp:36-41 / glm convertLinearToSRGB (parity unpinned: the
reference ships no picture fixtures). GPU part: trh_present_bgra8 through the C ABI against the
oracle on the SAME accumulated image: integer output, bit exact except where 255.99 * srgb(c)
lands within one float ulp of an integer (powf is libm on the CPU, ocml on the GPU)."""
import numpy as np
import pytest

from oracle.oracle import tonemap_bgra8


def test_oracle_tonemap_known_values():
    img = np.zeros((2, 3, 3), dtype=np.float32)
    img[0, 0] = [0.0, 1.0, 2.0]          # bottom-left: black / full / over-range
    img[1, 2] = [0.5, 0.002, -1.0]       # top-right: pow branch / linear branch / negative
    out = tonemap_bgra8(img, mode=1)
    assert out.shape == (2, 3)
    # row 0 of the picture is the TOP row: image row 1
    r, g, b = (out[0, 2] >> 16) & 255, (out[0, 2] >> 8) & 255, out[0, 2] & 255
    assert r == int(255.99 * (0.5 ** 0.41666 * 1.055 - 0.055))     # 187
    assert g == int(255.99 * 0.002 * 12.92)                        # 6
    assert b == 0
    assert out[1, 0] == (0 << 16) | (255 << 8) | 255
    assert (out >> 24 == 0).all()
    # Reinhard: x / (1 + x); ACES: the fitted rational curve; both before srgb()
    x = np.float32(1.0)
    out_r = tonemap_bgra8(np.full((1, 1, 3), x, np.float32), mode=2)
    assert (out_r[0, 0] & 255) == int(255.99 * (0.5 ** 0.41666 * 1.055 - 0.055))
    aces = (1.0 * (2.51 + 0.03)) / (1.0 * (2.43 + 0.59) + 0.14)
    out_a = tonemap_bgra8(np.full((1, 1, 3), x, np.float32), mode=3)
    assert (out_a[0, 0] & 255) == int(255.99 * (aces ** 0.41666 * 1.055 - 0.055))
    # no tonemap: clamp(255.99 * c), WITHOUT the exposure (the #else branch of cpu_details.cpp:236-238)
    out_n = tonemap_bgra8(np.full((1, 1, 3), 0.25, np.float32), mode=0, exposure=2.0)
    assert (out_n[0, 0] & 255) == int(255.99 * 0.25)
    out_s = tonemap_bgra8(np.full((1, 1, 3), 0.25, np.float32), mode=1, exposure=2.0)
    assert (out_s[0, 0] & 255) == int(255.99 * (0.5 ** 0.41666 * 1.055 - 0.055))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_present_matches_oracle_tonemap(scene_loader, mode):
    from tracy_amd import HipTrace
    sc = scene_loader("spheres", 160, 90)    # HDR sky: values far above 1 exercise the curves
    mod = HipTrace()
    mod.Startup(sc)
    try:
        mod.render(0, 8)
        img, n = mod.resolve()
        got = mod.present(tonemap=mode, exposure=1.5)
        want = tonemap_bgra8(img[..., :3], mode=mode, exposure=1.5)
        assert got.shape == want.shape and got.dtype == np.uint32
        d = np.zeros(got.shape, dtype=np.int64)
        for sh in (0, 8, 16):
            d = np.maximum(d, np.abs(((got >> sh) & 255).astype(np.int64) - ((want >> sh) & 255).astype(np.int64)))
        assert d.max() <= 1
        assert (d > 0).mean() < 2e-3
        assert (got >> 24 == 0).all()
    finally:
        mod.Shutdown()


@pytest.mark.gpu
def test_present_rejects_bad_arguments(scene_loader):
    import ctypes as C
    from tracy_amd import HipTrace, capi
    lib = capi.hip_lib()
    assert lib.trh_present_bgra8(None, 1, 1.0, None) != 0
    sc = scene_loader("testtree", 32, 18)
    mod = HipTrace()
    mod.Startup(sc)
    try:
        assert lib.trh_present_bgra8(mod._ctx, 7, C.c_float(1.0), None) != 0
        assert b"tonemap" in lib.trh_last_error()
        assert lib.trh_present_bgra8(mod._ctx, 1, C.c_float(1.0), None) == 0   # device-only present
    finally:
        mod.Shutdown()
