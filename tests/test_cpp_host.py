"""C++ host side (csrc/host/hip_trace.h, the HipTrace : TracyModule twin, and the headless CLI)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "tracy_amd", "lib", "tracy_hip_cli")


def test_generic_flatten_type_checks_against_reference_accessors(tmp_path):
    src = os.path.join(ROOT, "tests", "cpp", "flatten_reference_api_check.cpp")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-fsyntax-only", src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_cli_is_built_and_rejects_other_kernels():
    assert os.path.exists(CLI), "run __graft_entry__.build()"
    r = subprocess.run([CLI, "-kernel", "CUDA"], capture_output=True, text=True)
    assert r.returncode != 0 and "Invalid kernel choice" in r.stderr


def test_cli_reports_missing_scene():
    r = subprocess.run([CLI, "-scene", "does/not/exist.scn"], capture_output=True, text=True)
    assert r.returncode != 0 and "Unable to load scene" in r.stderr


def test_x11_viewer_is_built_and_fails_cleanly_without_a_display():
    """tracy_hip_view (the interactive twin of win_raytracer.cpp's loop) links against X11 and the two libraries;
    the boxes have no X server, so what can be checked is that it says so and exits before touching the GPU."""
    view = os.path.join(ROOT, "tracy_amd", "lib", "tracy_hip_view")
    if not os.path.exists("/usr/include/X11/Xlib.h"):
        pytest.skip("no X11 development files: the viewer is not built")
    assert os.path.exists(view), "run __graft_entry__.build()"
    env = {k: v for k, v in os.environ.items() if k != "DISPLAY"}
    r = subprocess.run([view, "-scene", os.path.join(ROOT, "assets/data/scenes/cornell.scn"), "-kernel", "hip"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode != 0 and "cannot open display" in r.stderr
    r = subprocess.run([view, "-kernel", "CUDA"], capture_output=True, text=True, env=env, timeout=60)
    assert r.returncode != 0 and "Invalid kernel choice" in r.stderr


@pytest.mark.gpu
def test_cli_renders_like_the_python_host(tmp_path):
    out = str(tmp_path / "cornell.pfm")
    r = subprocess.run([CLI, "-scene", os.path.join(ROOT, "assets/data/scenes/cornell.scn"), "-assets", os.path.join(ROOT, "assets"),
                        "-kernel", "hip", "-width", "96", "-height", "54", "-spp", "8", "-spu", "4", "-out", out],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "MRays/s" in r.stderr and "fps on average" in r.stderr
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        f.readline()
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    from tracy_amd import Scene, HipTrace
    sc = Scene("data/scenes/cornell.scn", 96, 54)
    mod = HipTrace()
    mod.Startup(sc)
    mod.render(0, 4)
    mod.render(4, 4)
    ref, _ = mod.resolve()
    mod.Shutdown()
    assert np.array_equal(img, ref[..., :3])


@pytest.mark.gpu
def test_cli_writes_the_tonemapped_picture_as_png_and_ppm(tmp_path):
    """-out x.png / x.ppm hold the module's own picture (trh_present_bgra8); the PNG must decode
    (PIL checks the chunk CRCs and the zlib adler32) to the same pixels as the PPM and as the
    Python host's present() of the same samples."""
    from PIL import Image
    from tracy_amd import Scene, HipTrace, capi
    common = [CLI, "-scene", os.path.join(ROOT, "assets/data/scenes/spheres.scn"), "-assets", os.path.join(ROOT, "assets"),
              "-kernel", "hip", "-width", "96", "-height", "54", "-spp", "8", "-spu", "8", "-tonemap", "aces", "-exposure", "1.25"]
    png, ppm = str(tmp_path / "a.png"), str(tmp_path / "a.ppm")
    for out in (png, ppm):
        r = subprocess.run(common + ["-out", out], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr
    a = np.asarray(Image.open(png).convert("RGB"))
    b = np.asarray(Image.open(ppm).convert("RGB"))
    assert a.shape == (54, 96, 3) and np.array_equal(a, b)
    sc = Scene("data/scenes/spheres.scn", 96, 54)
    mod = HipTrace()
    mod.Startup(sc)
    mod.render(0, 8)
    pic = mod.present(tonemap=capi.TRH_TONEMAP_ACES, exposure=1.25)
    mod.Shutdown()
    rgb = np.stack([(pic >> 16) & 255, (pic >> 8) & 255, pic & 255], axis=-1).astype(np.uint8)
    assert np.array_equal(a, rgb)


@pytest.mark.gpu
def test_cli_renders_the_textured_helmet_with_the_builtin_decoders(tmp_path):
    """helmet.scn needs five JPEG maps and (missing) sky.hdr: the C++ host decodes them itself and
    must produce exactly the image of the Python host, which takes the same decoders."""
    out = str(tmp_path / "helmet.pfm")
    r = subprocess.run([CLI, "-scene", os.path.join(ROOT, "assets/data/scenes/helmet.scn"), "-assets", os.path.join(ROOT, "assets"),
                        "-kernel", "hip", "-width", "96", "-height", "54", "-spp", "4", "-spu", "4", "-out", out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        f.readline()
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    from tracy_amd import Scene, HipTrace
    sc = Scene("data/scenes/helmet.scn", 96, 54)
    mod = HipTrace()
    mod.Startup(sc)
    mod.render(0, 4)
    ref, _ = mod.resolve()
    mod.Shutdown()
    assert np.array_equal(img, ref[..., :3])


@pytest.mark.gpu
def test_cli_adaptive_updates_trace_the_same_samples(tmp_path):
    """-spu auto (the module default): one sample for the first update, then doubling towards ~12 ms per update;
    -async: OnUpdate only enqueues, OnRender waits; -gpus 0: every GPU of the node through trh_group_*. Whatever
    the split into updates and devices, the sample set is the same: the image equals one trh_render of the same
    samples up to the rounding of the per-pixel float sums."""
    import re
    out = str(tmp_path / "dragon.pfm")
    r = subprocess.run([CLI, "-scene", os.path.join(ROOT, "assets/data/scenes/dragon.scn"), "-assets", os.path.join(ROOT, "assets"),
                        "-kernel", "hip", "-width", "320", "-height", "180", "-spp", "300", "-spu", "auto", "-async", "-gpus", "0", "-out", out],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    m = re.search(r"Msamples/s over (\d+) updates \((\d+) spp, last update (\d+) spp\)", r.stderr)
    assert m, r.stderr
    updates, spp, last = map(int, m.groups())
    assert spp >= 300 and last > 1 and updates < spp, "the samples per update never grew"
    with open(out, "rb") as f:
        assert f.readline() == b"PF\n"
        w, h = map(int, f.readline().split())
        f.readline()
        img = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    from tracy_amd import Scene, HipTrace
    sc = Scene("data/scenes/dragon.scn", 320, 180)
    mod = HipTrace()
    mod.Startup(sc)
    mod.render(0, spp)
    ref, _ = mod.resolve()
    mod.Shutdown()
    np.testing.assert_allclose(img, ref[..., :3], rtol=1e-5, atol=1e-6)
