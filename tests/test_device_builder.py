"""Device-side BVH build: the linear BVH built on the GPU (csrc/hip/lbvh_build.hip)
must be structurally sound (the host builder's own self-check, run on the arrays copied back), must
give the kernels the same nearest hits as the host-built SAH tree, and must be deterministic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MISS = 0xFFFFFFFF


@pytest.fixture(scope="module")
def ctx_mod(scene_loader):
    from tracy_amd import HipTrace
    mod = HipTrace()
    mod.Startup(scene_loader("cornell", 64, 36))
    yield mod
    mod.Shutdown()


@pytest.mark.parametrize("name", ["cornell", "dragon", "spheres", "random", "helmet", "bunny", "trimesh", "furnace"])
def test_structure_is_sound_and_deterministic(ctx_mod, scene_loader, name):
    sc = scene_loader(name, 64, 36)
    info, h1, bad = ctx_mod.bvh_build_device(sc)
    assert bad == 0, ctx_mod._lib.trh_last_error().decode()
    assert info["builder"] == 1 and info["tri_count"] == sc.tri_count
    assert 1 <= info["max_depth"] <= 32 and 1 <= info["max_leaf_size"] <= 4
    assert info["node_count"] < sc.tri_count and info["fp16_inflation"] >= 1.0
    _, h2, bad2 = ctx_mod.bvh_build_device(sc)
    assert bad2 == 0 and h2 == h1, "two builds of the same scene must give the same arrays"


@pytest.mark.parametrize("name,slack", [("dragon", 1.06), ("spheres", 1.10), ("helmet", 1.15), ("bunny", 1.08)])
def test_sah_cost_is_close_to_the_host_builders(ctx_mod, scene_loader, name, slack):
    """Round 3: SAH top tree over two-triangle Morton clusters + SAH splits below, all on the device. The tree must stay within
    a few per cent of the host SAH builder's cost (same C_traverse, same formula: trh_bvh_info.sah_cost); the plain linear BVH
    was 65 % (dragon) and 31 % (spheres) above it."""
    from tracy_amd import HipTrace
    sc = scene_loader(name, 64, 36)
    dev, _, bad = ctx_mod.bvh_build_device(sc)
    assert bad == 0, ctx_mod._lib.trh_last_error().decode()
    host = HipTrace()
    host.Startup(sc)
    try:
        ref = host.bvh_info()
    finally:
        host.Shutdown()
    assert ref["builder"] == 0 and dev["builder"] == 1
    print(f"\n{name}: SAH cost device {dev['sah_cost']:.2f}, host {ref['sah_cost']:.2f}, depth {dev['max_depth']} / {ref['max_depth']}")
    assert dev["sah_cost"] <= slack * ref["sah_cost"]


def test_subdivided_dragon_1p6m_triangles(ctx_mod):
    from tracy_amd import Scene
    sc = Scene("data/scenes/dragon.scn", 64, 36, subdivide=2)
    assert sc.tri_count > 1_500_000
    info, _, bad = ctx_mod.bvh_build_device(sc)
    assert bad == 0, ctx_mod._lib.trh_last_error().decode()
    print(f"\n1.6 M triangles: device build {info['device_build_ms']:.2f} ms (wall incl. upload {info['build_ms']:.1f} ms), "
          f"{info['node_count']} records, depth {info['max_depth']}, sah {info['sah_cost']:.1f}")
    sc.close()


@pytest.mark.parametrize("name,w,h", [("dragon", 320, 180), ("spheres", 192, 108), ("helmet", 192, 108), ("random", 192, 108), ("cornell", 192, 108)])
def test_renders_like_the_host_built_tree(scene_loader, oracle_mod, name, w, h):
    from tracy_amd import HipTrace, capi
    sc = scene_loader(name, w, h)
    a = HipTrace()
    a.Startup(sc)
    b = HipTrace(builder=capi.TRH_BUILDER_DEVICE_LBVH)
    b.Startup(sc)
    try:
        assert a.bvh_info()["builder"] == 0 and b.bvh_info()["builder"] == 1
        ia, ta, ua = a.primary_hits()
        ib, tb, ub = b.primary_hits()
        same = ia == ib
        assert same.mean() >= 0.9999, f"{(~same).sum()} primary-hit mismatches"
        ok = same & (ia != MISS)
        assert np.array_equal(ta[ok], tb[ok]) and np.array_equal(ua[ok], ub[ok])
        # and against the oracle directly
        ir, _, _ = oracle_mod.CpuRef(sc).primary_hits()
        assert (ib == ir).mean() >= 0.9999
        for s in (0, 7):
            ra, rb = a.sample_radiance(s), b.sample_radiance(s)
            close = np.isclose(ra[..., :3], rb[..., :3], rtol=1e-5, atol=1e-6).all(axis=-1)
            assert close.mean() >= 0.999, f"{name} sample {s}: {close.mean()}"
        a.render(0, 8)
        b.render(0, 8)
        ma, _ = a.resolve()
        mb, _ = b.resolve()
        assert np.isclose(ma[..., :3], mb[..., :3], rtol=1e-4, atol=1e-5).all(axis=-1).mean() >= 0.999
    finally:
        a.Shutdown()
        b.Shutdown()


def test_builder_default_from_the_environment(scene_loader, monkeypatch):
    """TRH_BUILDER=lbvh makes the device build the default of new contexts (hosts that cannot call trh_set_builder)."""
    from tracy_amd import HipTrace
    monkeypatch.setenv("TRH_BUILDER", "lbvh")
    mod = HipTrace()
    mod.Startup(scene_loader("bunny", 64, 36))
    try:
        assert mod.bvh_info()["builder"] == 1
    finally:
        mod.Shutdown()
    monkeypatch.delenv("TRH_BUILDER")
    mod = HipTrace()
    mod.Startup(scene_loader("bunny", 64, 36))
    try:
        assert mod.bvh_info()["builder"] == 0
    finally:
        mod.Shutdown()


def test_tiny_scenes_use_the_host_builder(scene_loader):
    from tracy_amd import HipTrace, capi
    mod = HipTrace(builder=capi.TRH_BUILDER_DEVICE_LBVH)
    mod.Startup(scene_loader("testtree", 64, 36))   # 4 triangles: one leaf
    try:
        assert mod.bvh_info()["builder"] == 0
        ids, _, _ = mod.primary_hits()
        assert (ids != MISS).any()
    finally:
        mod.Shutdown()
