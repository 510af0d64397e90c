"""HipTrace: the Python mirror of the `-kernel hip` render module.

Same life cycle, method names and argument meaning as the reference's TracyModule
(src/module.h:17-54) and its CPU implementation CpuTrace (cpu_trace.cpp:22-82); every method
is a thin call into the C ABI of include/tracy_hip.h. The C++ twin a Tracy maintainer would
compile into the application is csrc/host/hip_trace.h.

There is no fallback path: if libtracy_hip.so or a HIP device is missing, Startup raises.
"""
import ctypes as C

import numpy as np

from . import capi

DEFAULT_SEED = 0xABCDEF  # initrand(), src/random.h:66-69


class HipError(RuntimeError):
    pass


class HipTrace:
    # TracyEvent (src/module.h:12)
    eCreate, eResize, eCameraCut, eDestroy = range(4)

    def __init__(self, device=0, pipeline=None, spp_per_update=1, seed=DEFAULT_SEED, builder=None):
        self.builder = builder   # capi.TRH_BUILDER_*; None = the library default (host SAH)
        self._lib = None
        self._ctx = C.c_void_p()
        self.device = device
        self.pipeline = pipeline
        self.spp_per_update = spp_per_update
        self.seed = seed
        self._next_sample = 0
        self.width = self.height = 0

    def _check(self, rc):
        if rc:
            raise HipError(f"trh error {rc}: {self._lib.trh_last_error().decode()}")

    # -- TracyModule interface ----------------------------------------------------------
    def GetModuleName(self):
        return "HIP Raytracing [MI355X gfx950]"

    def Startup(self, scene, window=None):
        """bool Startup(const WindowHandle, const Scene&) (src/module.h:35)."""
        self._lib = capi.hip_lib()
        self._check(self._lib.trh_init(self.device, C.byref(self._ctx)))
        if self.builder is not None:
            self._check(self._lib.trh_set_builder(self._ctx, int(self.builder)))
        self._check(self._lib.trh_scene_upload(self._ctx, C.byref(scene.desc)))
        if self.pipeline is not None:
            self._check(self._lib.trh_set_pipeline(self._ctx, int(self.pipeline)))
        self.width, self.height = scene.width, scene.height
        self._upload_camera(scene)
        self._next_sample = 0
        return True

    def Shutdown(self):
        if self._lib is not None and self._ctx:
            self._lib.trh_shutdown(self._ctx)
            self._ctx = C.c_void_p()

    def OnUpdate(self, scene=None, dt=0.0):
        """One progressive pass: spp_per_update new samples for every pixel
        (CpuTrace::OnUpdate, cpu_trace.cpp:33-70, traces exactly one)."""
        self.render(self._next_sample, self.spp_per_update)
        self._next_sample += self.spp_per_update

    def OnEvent(self, event, scene=None, window=None):
        if event == self.eCameraCut:  # cpu_trace.cpp:76-78
            if scene is not None:
                self._upload_camera(scene)
            self._check(self._lib.trh_reset_accumulation(self._ctx))
            self._next_sample = 0

    def OnRender(self, window=None):
        """Returns the current linear mean image, (H, W, 4) float32, row 0 = bottom row."""
        return self.resolve()[0]

    def GetRayCount(self, reset=False):
        return self.stats(reset)["rays"]

    # -- direct C-ABI access used by tests and bench --------------------------------------
    def _upload_camera(self, scene):
        eye, ivp = scene.eye_ivp()
        self.set_camera(eye, ivp)

    def set_camera(self, eye, ivp):
        eye = np.ascontiguousarray(eye, dtype=np.float32)
        ivp = np.ascontiguousarray(ivp, dtype=np.float32)
        self._check(self._lib.trh_set_camera(self._ctx, eye.ctypes.data_as(C.POINTER(C.c_float)), ivp.ctypes.data_as(C.POINTER(C.c_float))))

    def set_debug_view(self, view, view3x3=None):
        """trh_set_debug_view: the reference's DEBUG_SHOW_* switches at run time (cpu_trace.cpp:127-137)."""
        v = None if view3x3 is None else np.ascontiguousarray(view3x3, dtype=np.float32).reshape(9)
        self._check(self._lib.trh_set_debug_view(self._ctx, int(view), None if v is None else v.ctypes.data))

    def set_pipeline(self, pipeline):
        self._check(self._lib.trh_set_pipeline(self._ctx, int(pipeline)))

    def reset(self):
        self._check(self._lib.trh_reset_accumulation(self._ctx))
        self._next_sample = 0

    def render(self, spp_begin, spp_count, seed=None, stats=False):
        fn = self._lib.trh_render_stats if stats else self._lib.trh_render
        self._check(fn(self._ctx, spp_begin, spp_count, self.seed if seed is None else seed))

    def render_noalloc(self, spp_begin, spp_count, seed=None):
        """trh_render_noalloc: the same launches with the launch scratch in the library's static device array."""
        self._check(self._lib.trh_render_noalloc(self._ctx, spp_begin, spp_count, self.seed if seed is None else seed))

    def synchronize(self):
        self._check(self._lib.trh_synchronize(self._ctx))

    def accum_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        self._check(self._lib.trh_accum_device_ptr(self._ctx, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_accum_buffer(self, dptr, nbytes):
        """Accumulate into caller-owned device memory (e.g. a torch tensor's data_ptr())."""
        self._check(self._lib.trh_bind_accum_buffer(self._ctx, dptr, nbytes))

    def resolve(self, to_host=True):
        n = C.c_uint64()
        if to_host:
            img = np.empty((self.height, self.width, 4), dtype=np.float32)
            self._check(self._lib.trh_resolve(self._ctx, img.ctypes.data_as(C.c_void_p), C.byref(n)))
            return img, n.value
        self._check(self._lib.trh_resolve(self._ctx, None, C.byref(n)))
        return None, n.value

    def present(self, tonemap=capi.TRH_TONEMAP_SRGB, exposure=1.0):
        """Tonemapped 0x00RRGGBB picture, row 0 = top (CPUDetails::UpdateBitmap, cpu_details.cpp:201-243)."""
        out = np.empty((self.height, self.width), dtype=np.uint32)
        self._check(self._lib.trh_present_bgra8(self._ctx, tonemap, exposure, out.ctypes.data_as(C.c_void_p)))
        return out

    def primary_hits(self):
        ids = np.empty((self.height, self.width), dtype=np.uint32)
        t = np.empty((self.height, self.width), dtype=np.float32)
        uv = np.empty((self.height, self.width, 2), dtype=np.float32)
        self._check(self._lib.trh_primary_hits(self._ctx, ids.ctypes.data_as(C.c_void_p), t.ctypes.data_as(C.c_void_p), uv.ctypes.data_as(C.c_void_p)))
        return ids, t, uv

    def sample_radiance(self, sample_index, seed=None):
        out = np.empty((self.height, self.width, 4), dtype=np.float32)
        self._check(self._lib.trh_sample_radiance(self._ctx, sample_index, self.seed if seed is None else seed, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self, reset=False):
        s = capi.trh_stats()
        self._check(self._lib.trh_get_stats(self._ctx, C.byref(s), int(reset)))
        return {k: getattr(s, k) for k, _ in capi.trh_stats._fields_}

    def debug_counters(self):
        """Wave-level iteration counts of the instrumented kernel (see trh_get_debug_counters)."""
        out = (C.c_uint64 * 19)()
        self._check(self._lib.trh_get_debug_counters(self._ctx, out, 19))
        return dict(zip(["wave_inner_iters", "wave_leaf_iters", "wave_shade_calls", "wave_gen_calls", "lanes_gen",
                         "cycles_inner", "cycles_leaf", "cycles_shade", "cycles_refill", "cycles_kernel", "cycles_longest_wave", "waves",
                         "t_first_start", "t_last_end", "t_sum_start", "t_sum_end",
                         "idle_leaf_in_inner", "idle_wait_in_inner", "idle_none_in_inner"], list(out)))

    def bvh_build_device(self, scene):
        """trh_bvh_build_device: (info dict, layout hash, violations) of the device-side linear BVH of `scene`."""
        b = capi.trh_bvh_info()
        h, v = C.c_uint64(0), C.c_uint64(0)
        self._check(self._lib.trh_bvh_build_device(self._ctx, C.byref(scene.desc), C.byref(b), C.byref(h), C.byref(v)))
        return {k: getattr(b, k) for k, _ in capi.trh_bvh_info._fields_}, h.value, v.value

    def bvh_info(self):
        b = capi.trh_bvh_info()
        self._check(self._lib.trh_get_bvh_info(self._ctx, C.byref(b)))
        return {k: getattr(b, k) for k, _ in capi.trh_bvh_info._fields_}

    def __del__(self):
        try:
            self.Shutdown()
        except Exception:
            pass


def split_samples(spp_begin, spp_count, ngpus, g):
    """trh_split_samples: the sample range device g of ngpus renders of a call of spp_count samples (host arithmetic)."""
    b, n = C.c_uint32(), C.c_uint32()
    capi.hip_lib().trh_split_samples(spp_begin, spp_count, ngpus, g, C.byref(b), C.byref(n))
    return b.value, n.value


class HipTraceGroup:
    """The module over the N GPUs of a node, driven by one host thread (trh_group_*, include/tracy_hip.h): the sample
    indices of every render call are split across the devices, OnRender / resolve run the one RCCL reduce to device 0.
    Same life cycle as HipTrace / TracyModule."""
    eCreate, eResize, eCameraCut, eDestroy = range(4)

    def __init__(self, ngpus=0, spp_per_update=None, seed=DEFAULT_SEED, builder=None):
        self.ngpus_requested = ngpus
        self.builder = builder
        self.seed = seed
        self.spp_per_update = spp_per_update
        self._lib = None
        self._grp = C.c_void_p()
        self.ngpus = 0
        self._next_sample = 0
        self.width = self.height = 0

    def _check(self, rc):
        if rc:
            raise HipError(f"trh error {rc}: {self._lib.trh_last_error().decode()}")

    def GetModuleName(self):
        return f"HIP Raytracing [{self.ngpus} x MI355X gfx950]"

    def Startup(self, scene, window=None):
        self._lib = capi.hip_lib()
        n = C.c_int(0)
        self._check(self._lib.trh_group_init(self.ngpus_requested, C.byref(n), C.byref(self._grp)))
        self.ngpus = n.value
        if self.spp_per_update is None:
            self.spp_per_update = self.ngpus   # one sample per device and update
        if self.builder is not None:
            self._check(self._lib.trh_group_set_builder(self._grp, int(self.builder)))
        self._check(self._lib.trh_group_scene_upload(self._grp, C.byref(scene.desc)))
        self.width, self.height = scene.width, scene.height
        self._upload_camera(scene)
        self._next_sample = 0
        return True

    def Shutdown(self):
        if self._lib is not None and self._grp:
            self._lib.trh_group_shutdown(self._grp)
            self._grp = C.c_void_p()

    def _upload_camera(self, scene):
        eye, ivp = scene.eye_ivp()
        eye = np.ascontiguousarray(eye, dtype=np.float32)
        ivp = np.ascontiguousarray(ivp, dtype=np.float32)
        self._check(self._lib.trh_group_set_camera(self._grp, eye.ctypes.data_as(C.POINTER(C.c_float)), ivp.ctypes.data_as(C.POINTER(C.c_float))))

    def OnUpdate(self, scene=None, dt=0.0):
        self.render(self._next_sample, self.spp_per_update)
        self._next_sample += self.spp_per_update

    def OnEvent(self, event, scene=None, window=None):
        if event == self.eCameraCut:
            if scene is not None:
                self._upload_camera(scene)
            self._check(self._lib.trh_group_reset_accumulation(self._grp))
            self._next_sample = 0

    def OnRender(self, window=None):
        return self.resolve()[0]

    def GetRayCount(self, reset=False):
        return self.stats(reset)["rays"]

    def render(self, spp_begin, spp_count, seed=None):
        self._check(self._lib.trh_group_render(self._grp, spp_begin, spp_count, self.seed if seed is None else seed))

    def synchronize(self):
        self._check(self._lib.trh_group_synchronize(self._grp))

    def reset(self):
        self._check(self._lib.trh_group_reset_accumulation(self._grp))
        self._next_sample = 0

    def resolve(self, to_host=True):
        n = C.c_uint64()
        img = np.empty((self.height, self.width, 4), dtype=np.float32) if to_host else None
        self._check(self._lib.trh_group_resolve(self._grp, img.ctypes.data_as(C.c_void_p) if to_host else None, C.byref(n)))
        return img, n.value

    def present(self, tonemap=capi.TRH_TONEMAP_SRGB, exposure=1.0):
        out = np.empty((self.height, self.width), dtype=np.uint32)
        self._check(self._lib.trh_group_present_bgra8(self._grp, tonemap, exposure, out.ctypes.data_as(C.c_void_p)))
        return out

    def stats(self, reset=False):
        s = capi.trh_stats()
        self._check(self._lib.trh_group_get_stats(self._grp, C.byref(s), int(reset)))
        return {k: getattr(s, k) for k, _ in capi.trh_stats._fields_}

    def __del__(self):
        try:
            self.Shutdown()
        except Exception:
            pass
