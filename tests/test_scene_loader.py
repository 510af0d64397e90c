"""Host scene loader (include/tracy_scene.h) against the facts the reference's scene files
and tessellators imply (src/scene.cpp:50-229; SURVEY.md section 8 sizes)."""
import numpy as np
import pytest


@pytest.mark.parametrize("name,tris,objects,materials", [
    ("cornell", 96, 8, 5),          # 8 boxes x 12 triangles
    ("spheres", 49600, 25, 26),     # 25 x (30*32*2 + 2*32) = 25 x 1984
    ("dragon", 100024, 3, 4),       # 100000 + 2 boxes
    ("random", 13973, 15, 10),      # 7 x 1984 + 7 x 12 + 1
    ("testtree", 4, 4, 5),
    ("furnace", 1984, 1, 2),
    ("bunny", 69666 + 24, 3, 4),    # stanford_bunny.obj (69 666 faces) + 2 boxes
    ("trimesh", 3 * 1984 + 12 + 6320, 5, 6),   # 3 spheres + 1 box + teapot.obj (quads fan-triangulated)
])
def test_scene_counts(scene_loader, name, tris, objects, materials):
    sc = scene_loader(name, 64, 36)
    assert sc.tri_count == tris
    assert sc.object_count == objects
    assert sc.desc.material_count == materials
    assert (sc.width, sc.height) == (64, 36)


def test_sphere_tessellation_quirks(scene_loader):
    sc = scene_loader("furnace", 64, 36)
    v = sc.vertices()
    assert v.shape == (4096, 14)            # 4 unshared vertices per quad (scene.cpp:79-96)
    n = v[:, 3:6]
    assert np.allclose(np.linalg.norm(n, axis=1), 1.0, atol=1e-6)
    assert np.allclose(v[:, 0:3], 2.0 * n, atol=1e-5)   # centre 0, radius 2
    m = sc.meshes()[0]
    assert np.allclose(m.aabb_min[:], [-2, -2, -2], atol=1e-2) and np.allclose(m.aabb_max[:], [2, 2, 2], atol=1e-2)


def test_box_layout(scene_loader):
    sc = scene_loader("cornell", 64, 36)
    meshes = sc.meshes()
    assert all(m.vertex_count == 24 and m.index_count == 36 for m in meshes)
    light = meshes[0]
    assert list(light.aabb_min[:]) == [213.0, 554.0, 227.0] and list(light.aabb_max[:]) == [343.0, 555.0, 332.0]
    mats = sc.materials()
    assert list(mats[1].emissive[:]) == [15.0, 15.0, 15.0]     # MTL Light (1,1,1) 0 0 0 15
    assert mats[0].emissive[0] == 0.0                           # no SKY line: black default material


def test_obj_import_dedupes_on_position_and_recomputes_flat_normals(scene_loader):
    sc = scene_loader("dragon", 64, 36)
    m = sc.meshes()[2]
    assert m.vertex_count == 50000 and m.index_count == 300000
    v = sc.vertices()[m.first_vertex:m.first_vertex + m.vertex_count]
    assert np.allclose(np.linalg.norm(v[:, 3:6], axis=1), 1.0, atol=1e-5)   # ComputeNormals (mesh.h:127-140)
    # last triangle wins: the normal stored on a vertex is the flat normal of the LAST triangle using it
    idx = sc.indices()[m.first_index:m.first_index + m.index_count].reshape(-1, 3)
    t = idx[-1]
    p = v[t, 0:3]
    fn = np.cross(p[1] - p[0], p[2] - p[0])
    fn /= np.linalg.norm(fn)
    assert np.allclose(v[t[0], 3:6], fn, atol=1e-4)


def test_camera_matrices(scene_loader):
    sc = scene_loader("dragon", 64, 36)
    cam = sc.camera()
    assert abs(cam.ratio - 64 / 36) < 1e-6
    P = np.array(cam.projection[:], dtype=np.float64).reshape(4, 4).T    # column-major -> math layout
    V = np.array(cam.view[:], dtype=np.float64).reshape(4, 4).T
    I = np.array(cam.inv_view_proj[:], dtype=np.float64).reshape(4, 4).T
    assert np.allclose(I @ (P @ V), np.eye(4), atol=1e-3)
    # eye maps to the view-space origin
    e = V @ np.array([*cam.eye[:], 1.0])
    assert np.allclose(e[:3], 0, atol=1e-5)
    # glm::perspective conventions (right handed, -1..1 clip)
    assert P[3, 2] == -1.0 and abs(P[1, 1] - 1 / np.tan(np.radians(60) / 2)) < 1e-6


def test_override_applies_before_cam(scene_loader):
    a = scene_loader("cornell", 0, 0)
    assert (a.width, a.height) == (800, 800) and abs(a.camera().ratio - 1.0) < 1e-7
    b = scene_loader("cornell", 1920, 1080)
    assert abs(b.camera().ratio - 1920 / 1080) < 1e-6


def test_synthetic_sky_is_used_for_missing_hdr(scene_loader):
    sc = scene_loader("spheres", 64, 36)
    assert sc.synthetic_textures == ["data/sky.hdr"]
    assert sc.desc.texture_count == 1
    t = sc.desc.textures[0]
    assert (t.width, t.height) == (1024, 512)
    px = np.ctypeslib.as_array(t.rgba, shape=(512, 1024, 4))
    assert np.isfinite(px).all() and px[..., :3].max() > 10.0 and px[..., :3].min() > 0.0   # HDR sun lobe present
    assert sc.materials()[0].tex[4] == 0


def test_textured_scene_loads(scene_loader):
    sc = scene_loader("helmet", 64, 36)
    assert sc.tri_count == 15452
    assert sc.desc.texture_count == 6
    mat = sc.materials()[1]
    assert all(mat.tex[i] != 0xFFFFFFFF for i in range(5))
    alb = sc.desc.textures[mat.tex[0]]
    assert (alb.width, alb.height) == (2048, 2048)
