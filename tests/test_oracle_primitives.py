"""Edge-case table for the oracle's RayTriangle / RayAABB restatements
(src/collision.h:33-74, :119-131): back-face cull, EPS edge rejection, t ties, slab rules."""
import ctypes as C

import numpy as np


def f3(*v):
    return np.array(v, dtype=np.float32)


def ray_tri(L, o, d, v0, v1, v2, tmax=np.float32(3.402823466e38)):
    t = C.c_float(tmax)
    uv = np.zeros(2, dtype=np.float32)
    hit = L.oref_ray_triangle(o.ctypes.data, d.ctypes.data, v0.ctypes.data, v1.ctypes.data, v2.ctypes.data, C.byref(t), uv.ctypes.data)
    return bool(hit), t.value, uv


V0, V1, V2 = f3(0, 0, 0), f3(1, 0, 0), f3(0, 1, 0)   # normal +z (counter-clockwise seen from +z)


def test_front_face_hit_and_barycentrics(oracle_mod):
    L = oracle_mod.lib()
    hit, t, uv = ray_tri(L, f3(0.25, 0.25, 1), f3(0, 0, -1), V0, V1, V2)
    assert hit and t == 1.0 and uv[0] == 0.25 and uv[1] == 0.25


def test_back_face_is_culled(oracle_mod):
    L = oracle_mod.lib()
    hit, _, _ = ray_tri(L, f3(0.25, 0.25, -1), f3(0, 0, 1), V0, V1, V2)
    assert not hit     # det < EPS (collision.h:44)


def test_parallel_ray_rejected(oracle_mod):
    L = oracle_mod.lib()
    hit, _, _ = ray_tri(L, f3(0.25, 0.25, 1), f3(1, 0, 0), V0, V1, V2)
    assert not hit


def test_edges_u0_v0_rejected_hypotenuse_accepted(oracle_mod):
    L = oracle_mod.lib()
    d = f3(0, 0, -1)
    assert not ray_tri(L, f3(0.0, 0.5, 1), d, V0, V1, V2)[0]   # u == 0 < EPS (collision.h:51)
    assert not ray_tri(L, f3(0.5, 0.0, 1), d, V0, V1, V2)[0]   # v == 0 < EPS (collision.h:59)
    assert ray_tri(L, f3(0.5, 0.5, 1), d, V0, V1, V2)[0]       # u + v == det is inside (collision.h:59)
    assert not ray_tri(L, f3(0.75, 0.5, 1), d, V0, V1, V2)[0]


def test_t_rules(oracle_mod):
    L = oracle_mod.lib()
    o, d = f3(0.25, 0.25, 1), f3(0, 0, -1)
    assert ray_tri(L, o, d, V0, V1, V2, tmax=np.float32(1.0))[0]       # equal t replaces (t > RayT rejects only)
    assert not ray_tri(L, o, d, V0, V1, V2, tmax=np.float32(0.999))[0]
    assert not ray_tri(L, f3(0.25, 0.25, -1), f3(0, 0, -1), V0, V1, V2)[0]  # behind the origin: t < EPS


def ray_box(L, o, d, bmin, bmax, tmax=3.0e38):
    return bool(L.oref_ray_aabb(o.ctypes.data, d.ctypes.data, bmin.ctypes.data, bmax.ctypes.data, C.c_float(tmax)))


def test_ray_aabb(oracle_mod):
    L = oracle_mod.lib()
    bmin, bmax = f3(-1, -1, -1), f3(1, 1, 1)
    assert ray_box(L, f3(0, 0, 5), f3(0, 0, -1), bmin, bmax)
    assert not ray_box(L, f3(0, 0, 5), f3(0, 0, 1), bmin, bmax)          # box behind
    assert not ray_box(L, f3(3, 0, 5), f3(0, 0, -1), bmin, bmax)         # passes beside (inf slabs)
    assert ray_box(L, f3(0, 0, 0), f3(0, 0, -1), bmin, bmax)             # origin inside: tmax >= EPS
    assert not ray_box(L, f3(0, 0, 5), f3(0, 0, -1), bmin, bmax, tmax=4.0)   # tmin (4) < tmax is strict
    assert ray_box(L, f3(0, 0, 5), f3(0, 0, -1), bmin, bmax, tmax=4.0001)
    # flat box (degenerate in y) is still hit
    assert ray_box(L, f3(0, 5, 0), f3(0, -1, 0), f3(-1, 0, -1), f3(1, 0, 1))
