import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
SCENES = ["cornell", "dragon", "spheres", "random", "testtree", "furnace"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.lib()
    return oracle


_scene_cache = {}


def load_scene(name, w, h):
    from tracy_amd.scene import Scene
    key = (name, w, h)
    if key not in _scene_cache:
        _scene_cache[key] = Scene(f"data/scenes/{name}.scn", w, h)
    return _scene_cache[key]


@pytest.fixture(scope="session")
def scene_loader():
    return load_scene


@pytest.fixture(scope="session")
def gpu_available():
    try:
        import ctypes as C
        from tracy_amd import capi
        n = C.c_int(0)
        rc = capi.hip_lib().trh_device_count(C.byref(n))
        return rc == 0 and n.value > 0
    except Exception:
        return False
