"""Debug views: the reference's compile-time DEBUG_SHOW_* switches (CMakeLists.txt:23-35,
cpu_trace.cpp:127-137) as a run-time setting of the module (trh_set_debug_view).

CPU part: the oracle's restatement against values that can be read off the scene files.
GPU part: the HIP module against the oracle, per sample (the views are deterministic first-hit
images, so the bar is equality up to the libm/ocml ulps of the normalisation, on every pixel
whose primary hit is the same triangle)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tracy_amd", "lib", "tracy_hip_cli")
VIEWS = {"basecolor": 1, "normals": 2, "metalness": 3, "roughness": 4, "emissive": 5}


def test_oracle_views_return_the_scene_files_material_inputs(oracle_mod, scene_loader):
    sc = scene_loader("cornell", 64, 36)
    ref = oracle_mod.CpuRef(sc)
    mats = sc.materials()
    radiance = ref.sample_radiance(0)
    for name, view in VIEWS.items():
        ref.set_debug_view(view, sc.view3x3())
        img = ref.sample_radiance(0)
        assert (img[..., 3] == 1).all(), "a debug path ends at its first bounce iteration"
        px = img[..., :3].reshape(-1, 3)
        if name == "basecolor":
            allowed = np.array([m.albedo[:] for m in mats] + [mats[0].emissive[:]], dtype=np.float32)
        elif name == "emissive":
            allowed = np.array([m.emissive[:] for m in mats], dtype=np.float32)
        elif name == "metalness":
            allowed = np.array([[m.metalness] * 3 for m in mats] + [mats[0].emissive[:]], dtype=np.float32)
        elif name == "roughness":
            allowed = np.array([[m.roughness] * 3 for m in mats] + [mats[0].emissive[:]], dtype=np.float32)
        else:
            # .5 * normalize(1 + n_view): a vector of length one half; paths that miss return the (black) sky
            nrm = np.linalg.norm(px, axis=1)
            assert (np.isclose(nrm, 0.5, atol=1e-6) | (nrm == 0)).all() and np.isclose(nrm, 0.5, atol=1e-6).mean() > 0.5
            # the camera looks down -z at the back wall, whose view-space normal is +z: 0.5 * (1,1,2)/sqrt(6)
            back = np.isclose(px, 0.5 * np.array([1, 1, 2]) / np.sqrt(6), atol=1e-5).all(axis=1)
            assert back.mean() > 0.05
            continue
        d = np.abs(px[:, None, :] - allowed[None, :, :]).max(axis=2).min(axis=1)
        assert (d == 0).all(), f"{name}: a pixel holds a value no material has"
    ref.set_debug_view(0)
    assert np.array_equal(ref.sample_radiance(0), radiance)


def test_oracle_textured_views_read_the_maps(oracle_mod, scene_loader):
    """helmet: every map is used; the base-colour view of a hit pixel is a texel of the base-colour map"""
    sc = scene_loader("helmet", 64, 36)
    ref = oracle_mod.CpuRef(sc)
    ref.set_debug_view(VIEWS["basecolor"], sc.view3x3())
    a = ref.sample_radiance(0)
    ref.set_debug_view(VIEWS["roughness"], sc.view3x3())
    b = ref.sample_radiance(0)
    ids, _, _ = ref.primary_hits()
    hit = ids != 0xFFFFFFFF
    assert hit.sum() > 200
    assert len(np.unique(a[hit][:, :3], axis=0)) > 50 and len(np.unique(b[hit][:, 0])) > 20
    # (the hit mask is of the un-jittered rays: silhouette pixels of the jittered sample may show the sky instead)
    grey = (b[hit][:, 0] == b[hit][:, 1]) & (b[hit][:, 0] == b[hit][:, 2])
    assert grey.mean() > 0.9


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cornell", "spheres", "helmet", "random", "dragon"])
def test_views_match_the_oracle_per_sample(oracle_mod, scene_loader, name):
    from tracy_amd import HipTrace
    sc = scene_loader(name, 96, 54)
    ref = oracle_mod.CpuRef(sc)
    mod = HipTrace()
    mod.Startup(sc)
    try:
        for vname, view in VIEWS.items():
            ref.set_debug_view(view, sc.view3x3())
            mod.set_debug_view(view, sc.view3x3())
            for s in (0, 3):
                r, g = ref.sample_radiance(s), mod.sample_radiance(s)
                assert (g[..., 3] == r[..., 3]).all()
                close = np.isclose(g[..., :3], r[..., :3], rtol=2e-6, atol=1e-7).all(axis=-1)
                # the few pixels that may differ: jittered primary rays whose nearest hit is a tie (see parity (a))
                assert close.mean() >= 0.999, f"{name} {vname} sample {s}: {close.mean()}"
        # the accumulating path (megakernel and wavefront) shows the same view
        mod.set_debug_view(VIEWS["basecolor"], sc.view3x3())
        want = (mod.sample_radiance(0)[..., :3].astype(np.float64) + mod.sample_radiance(1)[..., :3]) / 2
        for pipeline in (0, 1):
            mod.set_pipeline(pipeline)
            mod.reset()
            mod.render(0, 2)
            img, n = mod.resolve()
            assert n == 2 and np.allclose(img[..., :3], want, rtol=1e-6, atol=1e-7)
        # and switching the view off brings the radiance back
        mod.set_pipeline(0)
        mod.set_debug_view(0)
        ref.set_debug_view(0)
        r, g = ref.sample_radiance(0), mod.sample_radiance(0)
        assert np.isclose(g[..., :3], r[..., :3], rtol=1e-4, atol=1e-5).all(axis=-1).mean() >= 0.995
    finally:
        mod.Shutdown()


@pytest.mark.gpu
def test_cli_debug_flag(tmp_path):
    """-debug normals through the C++ module (HipTrace::SetDebugView) == the Python host's view"""
    from tracy_amd import Scene, HipTrace
    out = str(tmp_path / "n.pfm")
    r = subprocess.run([CLI, "-scene", os.path.join(ROOT, "assets/data/scenes/spheres.scn"), "-assets", os.path.join(ROOT, "assets"),
                        "-kernel", "hip", "-width", "96", "-height", "54", "-spp", "2", "-spu", "2", "-debug", "normals", "-out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        f.readline()
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    sc = Scene("data/scenes/spheres.scn", 96, 54)
    mod = HipTrace()
    mod.Startup(sc)
    mod.set_debug_view(VIEWS["normals"], sc.view3x3())
    mod.render(0, 2)
    want, _ = mod.resolve()
    mod.Shutdown()
    assert np.array_equal(img, want[..., :3])
