"""The integration artefacts a Tracy maintainer applies (integration/): the patches must apply to the reference's
own src/win_raytracer.cpp (:20-102: the `-kernel hip` case) and CMakeLists.txt (:169-172, :218-271: MODULE_HIP).
Build container only: the reference tree does not exist on the GPU box."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"

needs_reference = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "src", "win_raytracer.cpp")), reason="no reference tree here")


@needs_reference
def test_patches_apply_to_the_reference(tmp_path):
    os.makedirs(tmp_path / "src")
    shutil.copy(os.path.join(REF, "src", "win_raytracer.cpp"), tmp_path / "src" / "win_raytracer.cpp")
    shutil.copy(os.path.join(REF, "CMakeLists.txt"), tmp_path / "CMakeLists.txt")
    for patch in ("win_raytracer.patch", "CMakeLists.patch"):
        for extra in (["--dry-run"], []):
            r = subprocess.run(["patch", "-p1", "--forward", *extra, "-i", os.path.join(ROOT, "integration", patch)], cwd=tmp_path, capture_output=True, text=True)
            assert r.returncode == 0, patch + ": " + r.stdout + r.stderr
    main = (tmp_path / "src" / "win_raytracer.cpp").read_text()
    assert 'kernelname == "hip"' in main and "case TracyKernel::eHIP:" in main and "new HipTrace(TRACY_HIP_GPUS)" in main
    assert '#include "kernels/raytracing/hip/hip_module.h"' in main and "!defined(HIP_KERNEL)" in main
    cm = (tmp_path / "CMakeLists.txt").read_text()
    assert "option(MODULE_HIP" in cm and "include(./ext/tracy_hip/tracy_hip.cmake)" in cm
    # the block sits with the other module blocks, before the executable is declared
    assert cm.index("if(MODULE_HIP)") < cm.index("add_executable(")
    # ... and the executable waits for the library the fragment builds from source (a parallel build must not link before it exists)
    assert "add_dependencies(${PROJECT_NAME} tracy_hip_library)" in cm
    assert cm.index("add_executable(") < cm.index("add_dependencies(${PROJECT_NAME} tracy_hip_library)")


def test_cmake_fragment_and_adapter_name_existing_files():
    frag = open(os.path.join(ROOT, "integration", "tracy_hip.cmake")).read()
    for src in ("csrc/hip/tracy_hip.hip", "csrc/hip/tracy_group.hip", "csrc/hip/lbvh_build.hip", "csrc/hip/tile_order.hip", "csrc/host/bvh_builder.cpp"):
        assert src in frag and os.path.exists(os.path.join(ROOT, "tracy_amd", src)), src
    # the fragment and tracy_amd/build.py must list the same translation units
    build = open(os.path.join(ROOT, "tracy_amd", "build.py")).read()
    for unit in ("tracy_hip.hip", "tracy_group.hip", "lbvh_build.hip", "tile_order.hip", "bvh_builder.cpp"):
        assert unit in build
    assert "-DHIP_KERNEL=1" in frag and "--offload-arch=gfx950" in frag
    adapter = open(os.path.join(ROOT, "integration", "hip_module.h")).read()
    assert "tracy_hip::HipTraceT<Scene, TracyModule, WindowHandle, TracyEvent>" in adapter


def test_adapter_compiles_against_reference_shaped_types(tmp_path):
    """integration/hip_module.h compiled the way Tracy compiles it: after "module.h" and "scene.h", with
    TRACY_HIP_USE_REFERENCE_MODULE_H set, HipTrace deriving from the real-named TracyModule. The two headers here are
    test doubles with the reference's interface shapes (tests/cpp/reference_shapes.h; the real ones need the
    un-vendored cclib). A main() like win_raytracer.cpp's factory instantiates and drives the class (type check only)."""
    shapes = os.path.join(ROOT, "tests", "cpp", "reference_shapes.h")
    (tmp_path / "module.h").write_text(f'#pragma once\n#include "{shapes}"\n')
    (tmp_path / "scene.h").write_text(f'#pragma once\n#include "{shapes}"\n')
    src = tmp_path / "use.cpp"
    src.write_text('''#define TRACY_HIP_GPUS 0
#include "hip_module.h"
TracyModule* g_kernel{};
int main() {
    g_kernel = new HipTrace(TRACY_HIP_GPUS);           // InitializeKernel (win_raytracer.cpp:58-102)
    Scene world;
    handle_t win{640, 480, 0, nullptr, 0};
    if (!g_kernel->Startup(&win, world)) return 1;     // :480
    g_kernel->OnEvent(TracyEvent::eCameraCut, &win, world);   // :504
    g_kernel->OnUpdate(world, 0.f);                    // :511-515
    g_kernel->OnRender(&win);                          // :517
    const u32 rays = g_kernel->GetRayCount(true);      // :524
    g_kernel->Shutdown();                              // :558
    delete g_kernel;
    return rays ? 0 : 0;
}
''')
    # -fno-rtti -fno-exceptions: the reference's own flags (CMakeLists.txt:361)
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-fno-rtti", "-fno-exceptions", "-fsyntax-only", "-I", str(tmp_path),
                        "-I", os.path.join(ROOT, "integration"), "-I", os.path.join(ROOT, "tracy_amd", "csrc", "host"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
