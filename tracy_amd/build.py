"""Build recipes for the native libraries (in-tree, gfx950 only).

    python -m tracy_amd.build            # everything
    python -m tracy_amd.build hip scene cli

libtracy_scene.so : g++   host scene loader            (include/tracy_scene.h)
libtracy_hip.so   : hipcc kernels + C ABI + BVH build  (include/tracy_hip.h)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "lib")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-Wno-unused-function"]
HOST_FLAGS = ["-std=c++17", "-O2", "-fPIC", "-ffp-contract=off", "-Wall", "-Wextra"]


def _run(cmd):
    print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _glob_deps(*dirs):
    deps = []
    for d in dirs:
        for f in os.listdir(d):
            if f.endswith((".h", ".hip", ".cpp")):
                deps.append(os.path.join(d, f))
    return deps


def build_scene(force=False):
    out = os.path.join(LIB, "libtracy_scene.so")
    deps = _glob_deps(os.path.join(HERE, "csrc", "host"), os.path.join(ROOT, "include"))
    if force or _stale(out, deps):
        os.makedirs(LIB, exist_ok=True)
        _run(["g++", *HOST_FLAGS, "-shared", "-I", os.path.join(ROOT, "include"),
              os.path.join(HERE, "csrc", "host", "scene_host.cpp"), os.path.join(HERE, "csrc", "host", "image_decode.cpp"), "-o", out])
    return out


def build_hip(force=False, extra=()):
    out = os.path.join(LIB, "libtracy_hip.so")
    deps = _glob_deps(os.path.join(HERE, "csrc", "host"), os.path.join(HERE, "csrc", "hip"), os.path.join(ROOT, "include"))
    if force or _stale(out, deps):
        os.makedirs(LIB, exist_ok=True)
        extra = list(extra) + os.environ.get("TRH_HIPCC_EXTRA", "").split()
        _run([HIPCC, *HIP_FLAGS, *extra, "-shared",
              os.path.join(HERE, "csrc", "hip", "tracy_hip.hip"), os.path.join(HERE, "csrc", "hip", "lbvh_build.hip"),
              os.path.join(HERE, "csrc", "hip", "tile_order.hip"), os.path.join(HERE, "csrc", "hip", "tracy_group.hip"),
              os.path.join(HERE, "csrc", "host", "bvh_builder.cpp"), "-ldl", "-o", out])
    return out


def build_cli(force=False):
    """Headless C++ host (twin of the reference's main loop) linked against the two C-ABI libraries."""
    out = os.path.join(LIB, "tracy_hip_cli")
    deps = _glob_deps(os.path.join(HERE, "csrc", "host"), os.path.join(HERE, "csrc", "app"), os.path.join(ROOT, "include"))
    if force or _stale(out, deps):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", os.path.join(HERE, "csrc", "app", "tracy_hip_cli.cpp"),
              "-L", LIB, "-ltracy_hip", "-ltracy_scene", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib", "-o", out])
    return out


def build_view(force=False):
    """Interactive X11 host (csrc/app/tracy_hip_view.cpp). Built where the X11 development files exist; there is
    no X server on the GPU boxes, so only its start-up failure path is tested."""
    if not os.path.exists("/usr/include/X11/Xlib.h"):
        return None
    out = os.path.join(LIB, "tracy_hip_view")
    deps = _glob_deps(os.path.join(HERE, "csrc", "host"), os.path.join(HERE, "csrc", "app"), os.path.join(ROOT, "include"))
    if force or _stale(out, deps):
        _run(["g++", "-std=c++17", "-O2", "-Wall", "-Wextra", os.path.join(HERE, "csrc", "app", "tracy_hip_view.cpp"),
              "-L", LIB, "-ltracy_hip", "-ltracy_scene", "-lX11", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath-link," + "/opt/rocm/lib", "-o", out])
    return out


def build_all(force=False):
    build_scene(force)
    build_hip(force)
    build_cli(force)
    build_view(force)


if __name__ == "__main__":
    what = sys.argv[1:] or ["scene", "hip", "cli"]
    force = "--force" in what
    for w in what:
        if w == "scene":
            build_scene(force)
        elif w == "hip":
            build_hip(force)
        elif w == "cli":
            build_cli(force)
