"""Known-answer tests of the scene loader's tessellators, written out INDEPENDENTLY of the product's tables from
the reference source: Scene::AddBox (src/scene.cpp:133-214: 8 corners, 24 emitted vertices, 36 indices),
Scene::AddSphere (:50-131: 4 unshared vertices per quad, one triangle per cap quad), Scene::AddTriangle (:216-222),
Mesh::Transform / ComputeNormals (src/mesh.h:116-140) and the TRI transform order Rx Ry Rz T S (:478-483).

The oracle and the GPU module both consume the loader's arrays, so a swapped index or a wrong corner in the loader
would change both sides identically and pass every parity test; these tests are the loader's own pin (the
reference's OBJ and texture inputs are pinned by tests/test_reference_pins.py). Expected values are computed here
in float64 from the listing; positions produced with sin/cos are compared to 2e-6 of the radius (the reference's
sinf/cosf come from the missing cclib: SURVEY 8c), everything else exactly.
"""
import math

import numpy as np
import pytest

from tracy_amd.scene import Scene

HEAD = "SCN kat\nOUT 32 18\nCAM (0,0,10) (0,0,0) (0,1,0) 45\nMTL Grey (0.5,0.5,0.5) 1 0\n"


def load(tmp_path, body):
    p = tmp_path / "kat.scn"
    p.write_text(HEAD + body)
    return Scene(str(p))


# --- Scene::AddBox, read off src/scene.cpp:135-211 ---------------------------------------------------------------
def box_corners(b, t):
    # :135-144  { top.x, top.y, bottom.z } ... { bottom.x, bottom.y, top.z }
    return [(t[0], t[1], b[2]), (t[0], b[1], b[2]), (t[0], t[1], t[2]), (t[0], b[1], t[2]),
            (b[0], t[1], b[2]), (b[0], b[1], b[2]), (b[0], t[1], t[2]), (b[0], b[1], t[2])]


BOX_UV = [(0, 0), (1, 0), (0, 1), (1, 1)]                                             # :146-151
BOX_N = [(0, 1, 0), (0, 0, 1), (-1, 0, 0), (0, -1, 0), (1, 0, 0), (0, 0, -1)]          # :153-160
# emplace_back(vertices[c], normals[n], uv[u]) in emission order, :165-210
BOX_EMIT = [(4, 0, 0), (2, 0, 3), (0, 0, 1),   # 0 1 2      top      (+y)
            (2, 1, 1), (7, 1, 2), (3, 1, 3),   # 3 4 5      front    (+z)
            (6, 2, 1), (5, 2, 2), (7, 2, 3),   # 6 7 8      left     (-x)
            (1, 3, 3), (7, 3, 0), (5, 3, 2),   # 9 10 11    bottom   (-y)
            (0, 4, 1), (3, 4, 2), (1, 4, 3),   # 12 13 14   right    (+x)
            (4, 5, 1), (1, 5, 2), (5, 5, 3),   # 15 16 17   back     (-z)
            (6, 0, 2), (6, 1, 0), (4, 2, 0), (3, 3, 1), (2, 4, 0), (0, 5, 0)]   # 18 .. 23: the fourth corner of each face
BOX_INDICES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17,           # :168-193
               0, 18, 1, 3, 19, 4, 6, 20, 7, 9, 21, 10, 12, 22, 13, 15, 23, 16]        # :196-211


def test_box_vertices_and_indices_follow_the_reference_listing(tmp_path):
    b, t = (-1.0, -2.0, -3.0), (4.0, 5.0, 6.0)       # six different coordinates: every corner is identifiable
    sc = load(tmp_path, f"OBJ B ({b[0]},{b[1]},{b[2]}) ({t[0]},{t[1]},{t[2]}) Grey\n")
    v, idx, me = sc.vertices(), sc.indices(), sc.meshes()[0]
    assert (me.vertex_count, me.index_count, me.material) == (24, 36, 1)
    assert idx.tolist() == BOX_INDICES
    corners = box_corners(b, t)
    for k, (c, n, u) in enumerate(BOX_EMIT):
        assert tuple(v[k, 0:3]) == corners[c], f"vertex {k}: position"
        assert tuple(v[k, 3:6]) == BOX_N[n], f"vertex {k}: normal"          # Transform(identity) + normalize keep it
        assert tuple(v[k, 6:8]) == BOX_UV[u], f"vertex {k}: uv"
    assert tuple(me.aabb_min) == b and tuple(me.aabb_max) == t
    # every face is wound counter-clockwise seen from outside: RayTriangle culls back faces (collision.h:44), a
    # flipped triangle would be a hole
    for f in range(12):
        p0, p1, p2 = (v[idx[3 * f + k], 0:3].astype(np.float64) for k in range(3))
        n = np.cross(p1 - p0, p2 - p0)
        assert np.dot(n, v[idx[3 * f], 3:6]) > 0, f"triangle {f} faces inward"


def rot(axis, deg):
    a = math.radians(deg)
    c, s = math.cos(a), math.sin(a)
    if axis == 0:
        return np.array([[1, 0, 0, 0], [0, c, -s, 0], [0, s, c, 0], [0, 0, 0, 1.0]])
    if axis == 1:
        return np.array([[c, 0, s, 0], [0, 1, 0, 0], [-s, 0, c, 0], [0, 0, 0, 1.0]])
    return np.array([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]])


def translate(t):
    m = np.eye(4)
    m[:3, 3] = t
    return m


def test_rotated_box_uses_the_transform_and_normal_rule(tmp_path):
    """OBJ B min max (rot): rotation about the box centre, T(c) Rx Ry Rz T(-c) (src/scene.cpp:421-431); normals by
    normalize((transpose(inverse(M)) (n, 1)).xyz) (src/mesh.h:121)."""
    b, t, r = np.array([0.0, 0.0, 0.0]), np.array([2.0, 4.0, 6.0]), (30.0, 45.0, 60.0)
    sc = load(tmp_path, f"OBJ B (0,0,0) (2,4,6) ({r[0]},{r[1]},{r[2]}) Grey\n")
    v, idx = sc.vertices(), sc.indices()
    assert idx.tolist() == BOX_INDICES
    c = (b + t) / 2
    m = translate(c) @ rot(0, r[0]) @ rot(1, r[1]) @ rot(2, r[2]) @ translate(-c)
    nm = np.linalg.inv(m).T
    corners = box_corners(b, t)
    for k, (ci, n, u) in enumerate(BOX_EMIT):
        p = m @ np.array([*corners[ci], 1.0])
        q = nm @ np.array([*BOX_N[n], 1.0])
        q = q[:3] / np.linalg.norm(q[:3])
        np.testing.assert_allclose(v[k, 0:3], p[:3], rtol=0, atol=2e-6)
        np.testing.assert_allclose(v[k, 3:6], q, rtol=0, atol=2e-6)
        assert tuple(v[k, 6:8]) == BOX_UV[u]


# --- Scene::AddSphere, src/scene.cpp:50-131 ---------------------------------------------------------------------
def test_sphere_quads_caps_and_uv_follow_the_reference_listing(tmp_path):
    center, radius, steps = np.array([1.0, 2.0, -3.0]), 2.5, 32
    sc = load(tmp_path, "OBJ S (1,2,-3) 2.5 Grey\n")
    v, idx, me = sc.vertices(), sc.indices(), sc.meshes()[0]
    assert me.vertex_count == 4 * steps * steps           # four unshared vertices per quad
    assert me.index_count == 3 * (2 * steps * steps - 2 * steps)   # the two cap rows emit one triangle per quad
    want_idx, pos, uv = [], [], []
    for lon in range(steps):
        d1, d2 = lon / steps, (lon + 1) / steps
        th1, th2 = d1 * math.pi, d2 * math.pi
        for lat in range(steps):
            e1, e2 = lat / steps, (lat + 1) / steps
            ph1, ph2 = e1 * 2 * math.pi, e2 * 2 * math.pi
            for th, ph, du, dv in ((th1, ph1, e1, d1), (th1, ph2, e2, d1), (th2, ph2, e2, d2), (th2, ph1, e1, d2)):   # :79-96
                pos.append((math.sin(th) * math.cos(ph), math.sin(th) * math.sin(ph), math.cos(th)))
                uv.append((du, dv))
            base = len(pos) - 4
            if lon == 0:                      # :100-105 top cap
                want_idx += [base + 0, base + 3, base + 2]
            elif lon + 1 == steps:            # :106-111 end cap
                want_idx += [base + 2, base + 1, base + 0]
            else:                             # :112-121
                want_idx += [base + 0, base + 3, base + 1, base + 1, base + 3, base + 2]
    assert idx.tolist() == want_idx
    pos, uv = np.array(pos), np.array(uv)
    np.testing.assert_allclose(v[:, 0:3], center + pos * radius, rtol=0, atol=2e-6 * radius)
    np.testing.assert_allclose(v[:, 3:6], pos, rtol=0, atol=2e-6)       # normal = unit position (:81, :125-128)
    assert np.array_equal(v[:, 6:8], uv.astype(np.float32))             # uv = (phi / 2 pi, theta / pi), exact in float
    # first quad (top cap) and last quad (end cap), spelled out
    assert idx[:3].tolist() == [0, 3, 2] and idx[-3:].tolist() == [4 * steps * steps - 2, 4 * steps * steps - 3, 4 * steps * steps - 4]
    np.testing.assert_allclose(v[0, 0:3], center + np.array([0, 0, radius]), atol=1e-6)          # theta1 = 0: the pole
    np.testing.assert_allclose(v[-1, 0:3], center + np.array([0, 0, -radius]), atol=2e-6 * radius)   # theta2 = pi: the other pole
    # outward winding for every non-degenerate triangle (back faces are culled, collision.h:44)
    p = v[:, 0:3].astype(np.float64)
    tri = np.array(want_idx).reshape(-1, 3)
    n = np.cross(p[tri[:, 1]] - p[tri[:, 0]], p[tri[:, 2]] - p[tri[:, 0]])
    mid = (p[tri[:, 0]] + p[tri[:, 1]] + p[tri[:, 2]]) / 3 - center
    assert (np.einsum("ij,ij->i", n, mid) > 0).all()


# --- Scene::AddTriangle (:216-222), Mesh::ComputeNormals (mesh.h:127-140) ------------------------------------------
def test_triangle_gets_its_flat_normal(tmp_path):
    sc = load(tmp_path, "OBJ T (0,0,0) (2,0,0) (0,3,0) Grey\nOBJ T (1,1,1) (1,1,4) (1,5,1) Grey\n")
    v, idx = sc.vertices(), sc.indices()
    assert idx.tolist() == [0, 1, 2, 0, 1, 2]           # mesh-local indices
    assert [tuple(x) for x in v[0:3, 0:3]] == [(0, 0, 0), (2, 0, 0), (0, 3, 0)]
    assert [tuple(x) for x in v[0:3, 3:6]] == [(0, 0, 1)] * 3            # normalize(cross(v2 - v1, v3 - v1))
    assert [tuple(x) for x in v[3:6, 3:6]] == [(-1, 0, 0)] * 3           # cross((0,0,3), (0,4,0)) = (-12, 0, 0)


# --- TRI: transform = Rx Ry Rz T S (src/scene.cpp:478-483), position-only dedupe (:490-523), flat normals -----------
def test_obj_mesh_transform_order_dedupe_and_last_triangle_wins(tmp_path):
    obj = tmp_path / "two.obj"
    obj.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 0 0 1\nf 1 2 3\nf 1 3 4\n")
    tr, r, s = np.array([1.0, 2.0, 3.0]), (10.0, 20.0, 30.0), 2.0
    sc = load(tmp_path, f"TRI {obj} Grey ({tr[0]}, {tr[1]}, {tr[2]}) ({r[0]}, {r[1]}, {r[2]}) {s}\n")
    v, idx, me = sc.vertices(), sc.indices(), sc.meshes()[0]
    assert me.vertex_count == 4 and idx.tolist() == [0, 1, 2, 0, 2, 3]          # shared positions share a vertex
    m = rot(0, r[0]) @ rot(1, r[1]) @ rot(2, r[2]) @ translate(tr) @ np.diag([s, s, s, 1.0])
    src = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1.0]])
    want = (m @ np.c_[src, np.ones(4)].T).T[:, :3]
    np.testing.assert_allclose(v[:, 0:3], want, rtol=0, atol=2e-6)
    # no vn in the file: flat normals written onto the shared vertices, the LAST triangle that touches a vertex wins
    n1 = np.cross(want[1] - want[0], want[2] - want[0]); n1 /= np.linalg.norm(n1)
    n2 = np.cross(want[2] - want[0], want[3] - want[0]); n2 /= np.linalg.norm(n2)
    np.testing.assert_allclose(v[1, 3:6], n1, atol=2e-6)          # only in triangle 1
    for k in (0, 2, 3):
        np.testing.assert_allclose(v[k, 3:6], n2, atol=2e-6)      # triangle 2 wrote last


def test_a_swapped_index_would_be_noticed():
    """Guard of the guard: the expected tables above are not symmetric under the swaps a typo would make."""
    swapped = BOX_INDICES.copy()
    swapped[19], swapped[20] = swapped[20], swapped[19]
    assert swapped != BOX_INDICES and sorted(swapped) == sorted(BOX_INDICES)
