"""The C-ABI libraries load and export every function include/*.h declares.
No compute call is made here (there is no GPU in the CPU test run)."""
import ctypes as C
import os
import re

from tracy_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"_[a-z0-9_]+)\s*\(", txt)))


def test_hip_library_exports_every_declared_symbol():
    names = declared("tracy_hip.h", "trh")
    assert len(names) >= 18
    lib = capi.hip_lib()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(capi.HIP_SYMBOLS) == names
    assert lib.trh_abi_version() == 2


def test_scene_library_exports_every_declared_symbol():
    names = declared("tracy_scene.h", "tsc")
    lib = capi.scene_lib()
    for n in names:
        assert hasattr(lib, n), n
    assert sorted(capi.SCENE_SYMBOLS) == names


def test_struct_sizes_match_the_header():
    assert C.sizeof(capi.trh_material) == 60
    assert C.sizeof(capi.trh_mesh) == 44
    assert C.sizeof(capi.trh_texture) == 16
    assert C.sizeof(capi.trh_scene_desc) == 96
    assert C.sizeof(capi.trh_stats) == 64


def test_errors_are_reported_not_swallowed():
    lib = capi.hip_lib()
    assert lib.trh_render(None, 0, 1, 0) != 0
    assert b"NULL" in lib.trh_last_error()
    assert lib.trh_scene_upload(None, None) != 0
