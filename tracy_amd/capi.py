"""ctypes declarations of the two C ABIs (include/tracy_hip.h, include/tracy_scene.h).

Only declarations and library loading live here; no arithmetic. The HIP library is loaded
lazily and loading failures are raised, never papered over: there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
REPO_ROOT = os.path.dirname(_HERE)
ASSET_ROOT = os.path.join(REPO_ROOT, "assets")

TRH_NO_TEXTURE = 0xFFFFFFFF
TRH_VERTEX_FLOATS = 14
TRH_FLAG_RUSSIAN_ROULETTE = 1
TRH_PIPELINE_MEGAKERNEL = 0
TRH_PIPELINE_WAVEFRONT = 1
TRH_PIPELINE_MEGAKERNEL_SIMPLE = 2
TRH_PIPELINE_MEGAKERNEL_LANES = 3
TRH_TONEMAP_NONE, TRH_TONEMAP_SRGB, TRH_TONEMAP_REINHARD, TRH_TONEMAP_ACES = 0, 1, 2, 3
TRH_BUILDER_HOST_SAH, TRH_BUILDER_DEVICE_LBVH = 0, 1
TRH_VIEW_NONE, TRH_VIEW_BASECOLOR, TRH_VIEW_NORMALS, TRH_VIEW_METALNESS, TRH_VIEW_ROUGHNESS, TRH_VIEW_EMISSIVE = 0, 1, 2, 3, 4, 5


class trh_material(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("roughness", C.c_float), ("metalness", C.c_float), ("ior", C.c_float),
                ("emissive", C.c_float * 3), ("translucent", C.c_float), ("tex", C.c_uint32 * 5)]


class trh_texture(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("rgba", C.POINTER(C.c_float))]


class trh_mesh(C.Structure):
    _fields_ = [("first_vertex", C.c_uint32), ("vertex_count", C.c_uint32), ("first_index", C.c_uint32),
                ("index_count", C.c_uint32), ("material", C.c_uint32), ("aabb_min", C.c_float * 3), ("aabb_max", C.c_float * 3)]


class trh_scene_desc(C.Structure):
    _fields_ = [("vertices", C.POINTER(C.c_float)), ("vertex_count", C.c_uint32),
                ("indices", C.POINTER(C.c_uint32)), ("index_count", C.c_uint32),
                ("meshes", C.POINTER(trh_mesh)), ("mesh_count", C.c_uint32),
                ("materials", C.POINTER(trh_material)), ("material_count", C.c_uint32),
                ("textures", C.POINTER(trh_texture)), ("texture_count", C.c_uint32),
                ("width", C.c_uint32), ("height", C.c_uint32), ("max_bounces", C.c_uint32), ("flags", C.c_uint32)]


class trh_stats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("samples", C.c_uint64), ("box_tests", C.c_uint64), ("tri_tests", C.c_uint64),
                ("hits", C.c_uint64), ("tex_fetches", C.c_uint64), ("kernel_ms", C.c_double), ("launches", C.c_uint64)]


class trh_bvh_info(C.Structure):
    _fields_ = [("node_count", C.c_uint32), ("tri_count", C.c_uint32), ("max_depth", C.c_uint32), ("max_leaf_size", C.c_uint32),
                ("node_bytes", C.c_uint64), ("tri_bytes", C.c_uint64), ("sah_cost", C.c_double), ("build_ms", C.c_double),
                ("fp16_inflation", C.c_double), ("builder", C.c_uint32), ("from_cache", C.c_uint32), ("device_build_ms", C.c_double)]


class tsc_camera(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("center", C.c_float * 3), ("up", C.c_float * 3), ("fov_deg", C.c_float),
                ("ratio", C.c_float), ("view", C.c_float * 16), ("projection", C.c_float * 16), ("inv_view_proj", C.c_float * 16)]


class tsc_obj_counts(C.Structure):
    _fields_ = [("v_floats", C.c_uint32), ("vt_floats", C.c_uint32), ("vn_floats", C.c_uint32), ("shapes", C.c_uint32), ("corners", C.c_uint32)]


class tsc_texture_info(C.Structure):
    _fields_ = [("path", C.c_char_p), ("srgb", C.c_int), ("is_hdr", C.c_int), ("width", C.c_uint32), ("height", C.c_uint32)]


# every symbol include/tracy_scene.h declares: name -> (restype, argtypes)
SCENE_SYMBOLS = {
    "tsc_last_error": (C.c_char_p, []),
    "tsc_load": (C.c_void_p, [C.c_char_p, C.c_char_p, C.c_int, C.c_int]),
    "tsc_free": (None, [C.c_void_p]),
    "tsc_name": (C.c_char_p, [C.c_void_p]),
    "tsc_triangle_count": (C.c_uint32, [C.c_void_p]),
    "tsc_get_camera": (C.c_int, [C.c_void_p, C.POINTER(tsc_camera)]),
    "tsc_set_view": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "tsc_texture_count": (C.c_uint32, [C.c_void_p]),
    "tsc_get_texture_info": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(tsc_texture_info)]),
    "tsc_set_texture_u8": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "tsc_set_texture_f32": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]),
    "tsc_set_texture_synthetic_sky": (C.c_int, [C.c_void_p, C.c_uint32]),
    "tsc_decode_textures": (C.c_int, [C.c_void_p, C.c_int]),
    "tsc_decode_jpeg_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint8))]),
    "tsc_decode_hdr_file": (C.c_int, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float))]),
    "tsc_free_pixels": (None, [C.c_void_p]),
    "tsc_parse_obj_file": (C.c_int, [C.c_char_p, C.POINTER(tsc_obj_counts), C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)),
                                     C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.POINTER(C.c_int32))]),
    "tsc_subdivide": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "tsc_fill_desc": (C.c_int, [C.c_void_p, C.POINTER(trh_scene_desc)]),
}

# every symbol include/tracy_hip.h declares
HIP_SYMBOLS = {
    "trh_last_error": (C.c_char_p, []),
    "trh_abi_version": (C.c_int, []),
    "trh_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "trh_init": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "trh_shutdown": (C.c_int, [C.c_void_p]),
    "trh_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(trh_scene_desc)]),
    "trh_set_camera": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "trh_reset_accumulation": (C.c_int, [C.c_void_p]),
    "trh_set_pipeline": (C.c_int, [C.c_void_p, C.c_int]),
    "trh_set_debug_view": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "trh_render": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "trh_render_stats": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "trh_synchronize": (C.c_int, [C.c_void_p]),
    "trh_accum_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "trh_bind_accum_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "trh_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "trh_present_bgra8": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "trh_primary_hits": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "trh_sample_radiance": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.c_void_p]),
    "trh_get_stats": (C.c_int, [C.c_void_p, C.POINTER(trh_stats), C.c_int]),
    "trh_get_debug_counters": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_uint32]),
    "trh_bvh_build_host": (C.c_int, [C.POINTER(trh_scene_desc), C.POINTER(trh_bvh_info), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "trh_bvh_build_device": (C.c_int, [C.c_void_p, C.POINTER(trh_scene_desc), C.POINTER(trh_bvh_info), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "trh_set_builder": (C.c_int, [C.c_void_p, C.c_int]),
    "trh_set_bvh_cache_dir": (C.c_int, [C.c_char_p]),
    "trh_get_bvh_info": (C.c_int, [C.c_void_p, C.POINTER(trh_bvh_info)]),
    "trh_render_noalloc": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "trh_group_init": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "trh_group_shutdown": (C.c_int, [C.c_void_p]),
    "trh_group_size": (C.c_int, [C.c_void_p]),
    "trh_group_context": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "trh_group_set_builder": (C.c_int, [C.c_void_p, C.c_int]),
    "trh_group_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(trh_scene_desc)]),
    "trh_group_set_camera": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "trh_group_set_debug_view": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "trh_group_reset_accumulation": (C.c_int, [C.c_void_p]),
    "trh_group_render": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]),
    "trh_group_synchronize": (C.c_int, [C.c_void_p]),
    "trh_group_resolve": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "trh_group_present_bgra8": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_void_p]),
    "trh_group_get_stats": (C.c_int, [C.c_void_p, C.POINTER(trh_stats), C.c_int]),
    "trh_split_samples": (None, [C.c_uint32, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
}


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is missing: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


_scene_lib = None
_hip_lib = None


def scene_lib():
    global _scene_lib
    if _scene_lib is None:
        path = os.path.join(LIB_DIR, "libtracy_scene.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        _scene_lib = _bind(C.CDLL(path), SCENE_SYMBOLS)
    return _scene_lib


def hip_lib():
    """Load libtracy_hip.so (hand-written HIP kernels + C ABI). No fallback of any kind."""
    global _hip_lib
    if _hip_lib is None:
        path = os.environ.get("TRACY_HIP_LIBRARY") or os.path.join(LIB_DIR, "libtracy_hip.so")   # the override: A/B builds of the kernels
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: the HIP extension must be built (there is no CPU fallback)")
        _hip_lib = _bind(C.CDLL(path), HIP_SYMBOLS)
    return _hip_lib
