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
