"""ctypes face of oracle/liboracle_cpu_ref.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product package (tracy_amd/)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle_cpu_ref.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        L = C.CDLL(path)
        L.oref_create.restype = C.c_void_p
        L.oref_create.argtypes = [C.c_void_p]
        L.oref_destroy.argtypes = [C.c_void_p]
        L.oref_set_camera.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.oref_reset.argtypes = [C.c_void_p]
        L.oref_set_debug_view.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oref_set_matchance0.argtypes = [C.c_int]
        L.oref_set_matchance0.restype = None
        L.oref_tree_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.oref_primary_hits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oref_sample_radiance.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]
        L.oref_render.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_int]
        L.oref_render_serial_rng.argtypes = [C.c_void_p, C.c_uint32]
        L.oref_get_output.argtypes = [C.c_void_p, C.c_void_p]
        L.oref_get_counters.argtypes = [C.c_void_p] + [C.POINTER(C.c_uint64)] * 4 + [C.c_int]
        L.oref_fastrand.restype = C.c_float
        L.oref_fastrand.argtypes = [C.POINTER(C.c_uint32)]
        L.oref_path_rng_state.restype = C.c_uint32
        L.oref_path_rng_state.argtypes = [C.c_uint32, C.c_uint32, C.c_uint64]
        L.oref_ray_triangle.argtypes = [C.c_void_p] * 5 + [C.POINTER(C.c_float), C.c_void_p]
        L.oref_ray_aabb.argtypes = [C.c_void_p] * 4 + [C.c_float]
        L.oref_num_threads.restype = C.c_int
        L.oref_tonemap_bgra8.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_float, C.c_void_p]
        _lib = L
    return _lib


def tonemap_bgra8(rgb, mode=1, exposure=1.0):
    """CPUDetails::Tonemap + Bitmap::SetPixel on an (H, W, 3) linear image (row 0 = bottom)."""
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[:2]
    out = np.empty((h, w), dtype=np.uint32)
    lib().oref_tonemap_bgra8(rgb.ctypes.data, w, h, mode, exposure, out.ctypes.data)
    return out


class CpuRef:
    """The reference CPU module restated (CpuTrace, cpu_trace.cpp) over a trh_scene_desc."""

    def __init__(self, scene):
        self.L = lib()
        self.w, self.h = scene.width, scene.height
        self._scene = scene  # keeps the desc's buffers alive during oref_create
        self._h = self.L.oref_create(C.addressof(scene.desc))
        if not self._h:
            raise RuntimeError("oref_create failed")
        eye, ivp = scene.eye_ivp()
        self.set_camera(eye, ivp)

    def set_camera(self, eye, ivp):
        eye = np.ascontiguousarray(eye, dtype=np.float32)
        ivp = np.ascontiguousarray(ivp, dtype=np.float32)
        self.L.oref_set_camera(self._h, eye.ctypes.data, ivp.ctypes.data)

    def set_debug_view(self, view, view3x3=None):
        v = None if view3x3 is None else np.ascontiguousarray(view3x3, dtype=np.float32).reshape(9)
        self.L.oref_set_debug_view(self._h, int(view), None if v is None else v.ctypes.data)

    def reset(self):
        self.L.oref_reset(self._h)

    def tree_info(self):
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self.L.oref_tree_info(self._h, C.byref(a), C.byref(b), C.byref(c))
        return {"tlas_nodes": a.value, "blas_nodes": b.value, "blas_elements": c.value}

    def primary_hits(self):
        ids = np.empty((self.h, self.w), dtype=np.uint32)
        t = np.empty((self.h, self.w), dtype=np.float32)
        uv = np.empty((self.h, self.w, 2), dtype=np.float32)
        self.L.oref_primary_hits(self._h, ids.ctypes.data, t.ctypes.data, uv.ctypes.data)
        return ids, t, uv

    def sample_radiance(self, sample_index, seed=0xABCDEF):
        out = np.empty((self.h, self.w, 4), dtype=np.float32)
        self.L.oref_sample_radiance(self._h, sample_index, seed, out.ctypes.data)
        return out

    def trace_path(self, px, py, sample_index, seed=0xABCDEF, max_records=16):
        """[(hit id, t, rival id, rival t, candidates within 1e-4 t)] per bounce iteration of one (pixel, sample)."""
        rec = np.zeros((max_records, 6), dtype=np.float32)
        self.L.oref_trace_path.restype = C.c_uint32
        self.L.oref_trace_path.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_uint32]
        n = self.L.oref_trace_path(self._h, int(px), int(py), int(sample_index), int(seed), rec.ctypes.data, max_records)
        ids = rec.view(np.uint32)
        return [(int(ids[i, 0]), float(rec[i, 1]), int(ids[i, 2]), float(rec[i, 3]), int(rec[i, 4])) for i in range(min(n, max_records))]

    def render(self, spp_begin, spp_count, seed=0xABCDEF, count_traversal=False):
        self.L.oref_render(self._h, spp_begin, spp_count, seed, int(count_traversal))

    def render_serial_rng(self, frames):
        self.L.oref_render_serial_rng(self._h, frames)

    def output(self):
        out = np.empty((self.h, self.w, 3), dtype=np.float32)
        self.L.oref_get_output(self._h, out.ctypes.data)
        return out

    def counters(self, reset=False):
        v = [C.c_uint64() for _ in range(4)]
        self.L.oref_get_counters(self._h, *[C.byref(x) for x in v], int(reset))
        return {"rays": v[0].value, "frames": v[1].value, "box_tests": v[2].value, "tri_tests": v[3].value}

    def close(self):
        if self._h:
            self.L.oref_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
