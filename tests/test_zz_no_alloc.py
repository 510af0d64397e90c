"""trh_render_noalloc (include/tracy_hip.h): the entry point that allocates nothing. Its launch scratch lives in a
file-scope __device__ array of the library; the result must equal trh_render's on a context in the same state (one
without a tile order yet), on two calls in a row, the second with a smaller input -- the shared array then still
holds the first call's tile costs, order and sort scratch beyond the second call's sizes.
Bar: trh_render_noalloc launches the kernels of trh_render's TRH_PIPELINE_MEGAKERNEL_LANES pipeline with the same
arguments in the same order, so against a context set to that pipeline the accumulation buffers are compared BIT FOR
BIT (float32 sums and sample counts) and the ray counters for equality. Against the default pipeline (same paths, a
pixel's float additions in another order: its ring of paths needs hipMalloc'ed memory) the tolerance is
rtol 2e-5 + atol 1e-6 on the float32 sums, with equal sample counts and equal ray counters."""
import numpy as np
import pytest

from conftest import load_scene

pytestmark = pytest.mark.gpu


def _differing(a, b):
    """number of float32 words whose bits differ"""
    return int((a.view(np.uint32) != b.view(np.uint32)).sum())


def _render(scene, spp_begin, spp, noalloc, pipeline=None):
    from tracy_amd import HipTrace
    mod = HipTrace(device=0, pipeline=pipeline)
    mod.Startup(scene)
    (mod.render_noalloc if noalloc else mod.render)(spp_begin, spp)
    img, n = mod.resolve()
    st = mod.stats()
    mod.Shutdown()
    return img, n, st


def test_noalloc_equals_render_twice_second_call_smaller(gpu_available):
    assert gpu_available, "needs a HIP device"
    big = load_scene("dragon", 640, 360)      # 64 spp -> 8x4-pixel tiles: 7200 tiles in the scratch
    small = load_scene("cornell", 200, 120)   # 24 spp -> fewer, larger tiles; the scratch keeps the dragon's leftovers
    from tracy_amd import HipTrace
    a = HipTrace(device=0)
    a.Startup(big)
    b = HipTrace(device=0)
    b.Startup(small)
    # two calls in a row through the new entry point, the second with the smaller input
    # (each context renders on its own stream and all contexts of a GPU share the scratch array: calls that overlap on
    # one GPU race, so the first is synchronised before the second is enqueued -- include/tracy_hip.h)
    a.render_noalloc(0, 64)
    a.synchronize()
    b.render_noalloc(3, 24)
    b.synchronize()
    img_a, n_a = a.resolve()
    img_b, n_b = b.resolve()
    st_a, st_b = a.stats(), b.stats()
    # a third call on the first context: unordered launch (< 8 spp), tile counter only
    a.render_noalloc(64, 4)
    img_a2, n_a2 = a.resolve()
    a.Shutdown()
    b.Shutdown()

    from tracy_amd import capi
    ref_a, rn_a, rst_a = _render(big, 0, 64, noalloc=False, pipeline=capi.TRH_PIPELINE_MEGAKERNEL_LANES)
    ref_b, rn_b, rst_b = _render(small, 3, 24, noalloc=False, pipeline=capi.TRH_PIPELINE_MEGAKERNEL_LANES)
    assert n_a == rn_a == 64 and n_b == rn_b == 24
    assert _differing(img_a, ref_a) == 0, "first call differs from trh_render"
    assert _differing(img_b, ref_b) == 0, "second (smaller) call differs from trh_render"
    assert st_a["rays"] == rst_a["rays"] and st_b["rays"] == rst_b["rays"]
    assert st_a["samples"] == rst_a["samples"] and st_b["samples"] == rst_b["samples"]
    assert st_a["launches"] == 0, "the no-alloc entry point records no launch timing"
    assert n_a2 == 68 and np.isfinite(img_a2).all()
    # ... and against the default pipeline: same paths, another summation order
    for img, scene, begin, spp, st in ((img_a, big, 0, 64, st_a), (img_b, small, 3, 24, st_b)):
        dflt, dn, dst = _render(scene, begin, spp, noalloc=False)
        assert dn == spp and dst["rays"] == st["rays"]
        assert np.array_equal(img[..., 3], dflt[..., 3])
        assert np.allclose(img[..., :3], dflt[..., :3], rtol=2e-5, atol=1e-6)


def test_noalloc_at_the_documented_upper_limit(gpu_available):
    """3840x2160 at 64 spp per call = 259 200 tiles of 8x4 pixels, just under the 262 144 the static scratch holds (the
    repo's own helmet workload): the tile sort's temporary storage must fit too. Bit for bit against trh_render."""
    assert gpu_available, "needs a HIP device"
    from tracy_amd import HipTrace, capi
    sc = load_scene("cornell", 3840, 2160)
    import ctypes
    query = capi.hip_lib().trh_debug_sort_temp_bytes
    query.restype, query.argtypes = ctypes.c_size_t, [ctypes.c_uint32]
    need = query(262144)
    assert need <= 4 << 20, f"the tile sort wants {need} bytes of temporary storage at 262144 tiles"
    a = HipTrace(device=0)
    a.Startup(sc)
    a.render_noalloc(0, 64)
    img_a, n_a = a.resolve()
    a.Shutdown()
    ref, rn, _ = _render(sc, 0, 64, noalloc=False, pipeline=capi.TRH_PIPELINE_MEGAKERNEL_LANES)
    assert n_a == rn == 64
    assert _differing(img_a, ref) == 0


def test_noalloc_rejects_what_does_not_fit(gpu_available):
    assert gpu_available, "needs a HIP device"
    from tracy_amd import HipTrace, HipError, capi
    sc = load_scene("cornell", 4096, 4096)    # 16.8 M pixels at 512 spp: 4-pixel tiles -> 4.2 M tiles > 262144
    mod = HipTrace(device=0)
    mod.Startup(sc)
    with pytest.raises(HipError, match="tiles"):
        mod.render_noalloc(0, 512)
    with pytest.raises(HipError, match="65536"):
        mod.render_noalloc(0, 70000)
    mod.set_pipeline(capi.TRH_PIPELINE_WAVEFRONT)
    with pytest.raises(HipError, match="megakernel"):
        mod.render_noalloc(0, 8)
    mod.Shutdown()
