"""Host-side Scene: Python face of include/tracy_scene.h (mirrors the accessors of the
reference's `Scene`, src/scene.h:36-60, as far as a render module reads them)."""
import ctypes as C
import os

import numpy as np

from . import capi


class SceneError(RuntimeError):
    pass


class Scene:
    """`Scene::Init` (src/scene.cpp:231) + texture decoding.

    width/height override the OUT line before the CAM line is evaluated (the camera ratio is
    fixed at CAM time, src/scene.cpp:290). Relative paths inside the .scn resolve against
    asset_root (the reference runs with its repository root as CWD).
    """

    def __init__(self, scn_path, width=0, height=0, asset_root=None, missing_hdr="synthetic", decoder="builtin", subdivide=0,
                 subdivide_min_triangles=1000):
        self._lib = capi.scene_lib()
        self.asset_root = asset_root or capi.ASSET_ROOT
        if not os.path.isabs(scn_path) and not os.path.exists(scn_path):
            scn_path = os.path.join(self.asset_root, scn_path)
        self.path = scn_path
        self._h = self._lib.tsc_load(scn_path.encode(), self.asset_root.encode(), int(width), int(height))
        if not self._h:
            raise SceneError(self._lib.tsc_last_error().decode())
        self.synthetic_textures = []
        self._decode_textures(missing_hdr, decoder)
        if subdivide:   # stress variant: 4^subdivide times the triangles of every large mesh, same surface
            if self._lib.tsc_subdivide(self._h, int(subdivide_min_triangles), int(subdivide)):
                raise SceneError(self._lib.tsc_last_error().decode())
        self._desc = capi.trh_scene_desc()
        rc = self._lib.tsc_fill_desc(self._h, C.byref(self._desc))
        if rc:
            raise SceneError(self._lib.tsc_last_error().decode())

    # -- textures -------------------------------------------------------------------------
    def _decode_textures(self, missing_hdr, decoder):
        """Image decoding (stb_image in the reference, src/scene.cpp:23-48). Default: the scene
        library's own JPEG / Radiance-HDR decoders (tsc_decode_textures), the same path the C++
        host takes. decoder="pil" decodes 8-bit images with PIL instead (formats the built-in
        decoders do not cover, e.g. PNG): expanded to RGBA8 like stbi_load(..., 4). A missing
        .hdr is replaced by the documented synthetic probe."""
        n = self._lib.tsc_texture_count(self._h)
        for i in range(n):
            info = capi.tsc_texture_info()
            self._lib.tsc_get_texture_info(self._h, i, C.byref(info))
            rel = info.path.decode()
            full = rel if os.path.isabs(rel) else os.path.join(self.asset_root, rel)
            if info.is_hdr and not os.path.exists(full):
                if missing_hdr != "synthetic":
                    raise SceneError(f"{rel} is missing")
                self.synthetic_textures.append(rel)
            elif decoder == "pil" and not info.is_hdr:
                from PIL import Image
                img = Image.open(full).convert("RGBA")
                a = np.ascontiguousarray(np.asarray(img, dtype=np.uint8))
                if self._lib.tsc_set_texture_u8(self._h, i, a.shape[1], a.shape[0], a.ctypes.data_as(C.c_void_p)):
                    raise SceneError(self._lib.tsc_last_error().decode())
        if self._lib.tsc_decode_textures(self._h, 1 if missing_hdr == "synthetic" else 0):
            raise SceneError(self._lib.tsc_last_error().decode())

    # -- accessors ------------------------------------------------------------------------
    @property
    def desc(self):
        return self._desc

    @property
    def width(self):
        return int(self._desc.width)

    @property
    def height(self):
        return int(self._desc.height)

    @property
    def name(self):
        return self._lib.tsc_name(self._h).decode()

    @property
    def tri_count(self):
        return int(self._lib.tsc_triangle_count(self._h))

    @property
    def object_count(self):
        return int(self._desc.mesh_count)

    def camera(self):
        cam = capi.tsc_camera()
        self._lib.tsc_get_camera(self._h, C.byref(cam))
        return cam

    def eye_ivp(self):
        cam = self.camera()
        return (np.array(cam.eye[:], dtype=np.float32), np.array(cam.inv_view_proj[:], dtype=np.float32))

    def view3x3(self):
        """mat3(Camera::GetView()), column-major, 9 floats (the NORMALS debug view needs it)."""
        v = np.array(self.camera().view[:], dtype=np.float32).reshape(4, 4)   # row i = column i
        return np.ascontiguousarray(v[:3, :3]).reshape(9)

    def set_view(self, eye, center, up):
        f3 = C.c_float * 3
        rc = self._lib.tsc_set_view(self._h, f3(*eye), f3(*center), f3(*up))
        if rc:
            raise SceneError(self._lib.tsc_last_error().decode())

    def vertices(self):
        d = self._desc
        return np.ctypeslib.as_array(d.vertices, shape=(d.vertex_count, capi.TRH_VERTEX_FLOATS))

    def indices(self):
        d = self._desc
        return np.ctypeslib.as_array(d.indices, shape=(d.index_count,))

    def meshes(self):
        d = self._desc
        return [d.meshes[i] for i in range(d.mesh_count)]

    def materials(self):
        d = self._desc
        return [d.materials[i] for i in range(d.material_count)]

    def close(self):
        if getattr(self, "_h", None):
            self._lib.tsc_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
