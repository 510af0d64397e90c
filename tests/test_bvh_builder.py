"""Host BVH builder (csrc/host/bvh_builder.cpp) through the host-only entry trh_bvh_build_host:
structural self-check on every scene, thread-count independence of the layout, the fp16 inflation
measure, and the degenerate inputs (empty scene, one triangle, coincident triangles)."""
import ctypes as C

import numpy as np
import pytest

from tracy_amd import capi

ALL = ["cornell", "dragon", "spheres", "random", "testtree", "furnace", "helmet", "bunny", "trimesh"]


def build_host(desc):
    lib = capi.hip_lib()
    info, h, bad = capi.trh_bvh_info(), C.c_uint64(), C.c_uint64()
    rc = lib.trh_bvh_build_host(C.byref(desc), C.byref(info), C.byref(h), C.byref(bad))
    assert rc == 0, lib.trh_last_error()
    return info, h.value, bad.value, lib.trh_last_error().decode()


@pytest.mark.parametrize("name", ALL)
def test_tree_is_sound(scene_loader, name):
    sc = scene_loader(name, 64, 36)
    info, _, bad, why = build_host(sc.desc)
    assert bad == 0, why
    assert info.tri_count == sc.tri_count
    assert 1 <= info.max_depth <= 32 and 1 <= info.max_leaf_size <= 8
    assert info.node_count >= 1 and info.fp16_inflation >= 1.0
    # shipped scenes keep their geometry around the centre of mass: fp16 records are in use
    assert info.node_bytes == 32 * info.node_count and info.fp16_inflation < 1.25


def test_layout_does_not_depend_on_builder_threads(scene_loader, monkeypatch):
    sc = scene_loader("dragon", 64, 36)
    hashes = {}
    for t in ("1", "2", "8"):
        monkeypatch.setenv("TRH_BUILD_THREADS", t)
        info, h, bad, why = build_host(sc.desc)
        assert bad == 0, why
        hashes[t] = (h, info.node_count, info.max_depth, round(info.sah_cost, 9))
    assert hashes["1"] == hashes["2"] == hashes["8"]


def test_subdivided_dragon_and_big_node_parallel_binning(monkeypatch):
    from tracy_amd import Scene
    sc = Scene("data/scenes/dragon.scn", 64, 36, subdivide=1)      # 400 024 triangles: the root is binned by several threads
    assert sc.tri_count == 400024
    out = {}
    for t in ("1", "8"):
        monkeypatch.setenv("TRH_BUILD_THREADS", t)
        info, h, bad, why = build_host(sc.desc)
        assert bad == 0, why
        out[t] = (h, info.node_count)
    assert out["1"] == out["8"]


def make_desc(vertices, indices):
    v = np.zeros((len(vertices), capi.TRH_VERTEX_FLOATS), dtype=np.float32)
    v[:, :3] = vertices
    v[:, 3:6] = (0, 0, 1)
    idx = np.asarray(indices, dtype=np.uint32)
    mesh = capi.trh_mesh(0, len(v), 0, len(idx), 1)
    mats = (capi.trh_material * 2)()
    for m in mats:
        for k in range(5):
            m.tex[k] = capi.TRH_NO_TEXTURE
    d = capi.trh_scene_desc()
    d.vertices = v.ctypes.data_as(C.POINTER(C.c_float)); d.vertex_count = len(v)
    d.indices = idx.ctypes.data_as(C.POINTER(C.c_uint32)); d.index_count = len(idx)
    d.meshes = C.pointer(mesh); d.mesh_count = 1
    d.materials = mats; d.material_count = 2
    d.width, d.height, d.max_bounces, d.flags = 8, 8, 5, 1
    return d, (v, idx, mesh, mats)


def test_degenerate_inputs():
    d, keep = make_desc([(0, 0, 0)], [])                              # no triangle at all
    d.mesh_count = 0
    info, _, bad, why = build_host(d)
    assert bad == 0 and info.tri_count == 0 and info.node_count == 1, why
    d, keep = make_desc([(0, 0, 0), (1, 0, 0), (0, 1, 0)], [0, 1, 2])  # one triangle: root record wraps one leaf
    info, _, bad, why = build_host(d)
    assert bad == 0 and info.tri_count == 1 and info.node_count == 1 and info.max_depth == 1, why
    tri = [(0, 0, 0), (1, 0, 0), (0, 1, 0)]
    d, keep = make_desc(tri, [0, 1, 2] * 100)                         # 100 coincident triangles: SAH cannot split, median must
    info, _, bad, why = build_host(d)
    assert bad == 0 and info.tri_count == 100 and info.max_leaf_size <= 8 and info.max_depth <= 32, why


def test_far_apart_detail_is_flagged_for_fp32_records():
    rng = np.random.default_rng(1)
    verts, idx = [], []
    for cx in (0.0, 500.0):
        for k in range(400):
            p = rng.normal(0, 0.05, 3) + (cx, 0, 0)
            base = len(verts)
            verts += [p, p + (0.004, 0, 0), p + (0, 0.004, 0)]
            idx += [base, base + 1, base + 2]
    d, keep = make_desc(np.array(verts), idx)
    info, _, bad, why = build_host(d)
    assert bad == 0, why
    assert info.fp16_inflation > 1.25 and info.node_bytes == 64 * info.node_count


def test_invalid_descriptions_are_rejected():
    lib = capi.hip_lib()
    bad = C.c_uint64()
    assert lib.trh_bvh_build_host(None, None, None, C.byref(bad)) != 0
    d, keep = make_desc([(0, 0, 0), (1, 0, 0), (0, 1, 0)], [0, 1, 7])   # vertex index out of range
    assert lib.trh_bvh_build_host(C.byref(d), None, None, C.byref(bad)) != 0
    assert b"out of range" in lib.trh_last_error()
    for poison in (np.nan, np.inf, -np.inf):       # NaN / inf positions: TRH_ERR_INVALID_ARG, never into the builders
        d, keep = make_desc([(0, 0, 0), (1, poison, 0), (0, 1, 0)], [0, 1, 2])
        assert lib.trh_bvh_build_host(C.byref(d), None, None, C.byref(bad)) == 1
        assert b"not finite" in lib.trh_last_error()


def test_on_disk_cache(scene_loader, tmp_path):
    """trh_set_bvh_cache_dir: a second build of the same scene is read from the file (same arrays, hash-equal), a
    different scene gets its own file, and a damaged or truncated file is ignored and rewritten, never trusted."""
    import glob
    import os
    lib = capi.hip_lib()
    sc, other = scene_loader("dragon", 64, 36), scene_loader("bunny", 64, 36)
    _, h_plain, bad, why = build_host(sc.desc)
    assert bad == 0, why
    try:
        assert lib.trh_set_bvh_cache_dir(str(tmp_path).encode()) == 0
        info1, h1, bad1, _ = build_host(sc.desc)
        files = glob.glob(str(tmp_path / "*.trhbvh"))
        assert info1.from_cache == 0 and bad1 == 0 and h1 == h_plain and len(files) == 1
        info2, h2, bad2, _ = build_host(sc.desc)
        assert info2.from_cache == 1 and bad2 == 0 and h2 == h_plain
        assert (info2.node_count, info2.max_depth, info2.max_leaf_size) == (info1.node_count, info1.max_depth, info1.max_leaf_size)
        assert info2.fp16_inflation == info1.fp16_inflation and info2.sah_cost == info1.sah_cost
        info3, _, bad3, _ = build_host(other.desc)
        assert info3.from_cache == 0 and bad3 == 0 and len(glob.glob(str(tmp_path / "*.trhbvh"))) == 2
        # flip one byte in the middle of the node records: checksum mismatch -> rebuilt and rewritten
        path = files[0]
        size = os.path.getsize(path)
        with open(path, "r+b") as f:
            f.seek(96 + 4000)
            b = f.read(1)
            f.seek(96 + 4000)
            f.write(bytes([b[0] ^ 0x40]))
        info4, h4, bad4, _ = build_host(sc.desc)
        assert info4.from_cache == 0 and bad4 == 0 and h4 == h_plain and os.path.getsize(path) == size
        assert build_host(sc.desc)[0].from_cache == 1
        # truncated file
        with open(path, "r+b") as f:
            f.truncate(size // 2)
        info5, h5, bad5, _ = build_host(sc.desc)
        assert info5.from_cache == 0 and bad5 == 0 and h5 == h_plain
        # a file with a valid checksum over WRONG links must not get through: swap two child links of the root record
        with open(path, "r+b") as f:
            raw = bytearray(f.read())
        n_nodes_f32 = int.from_bytes(raw[16:24], "little")
        root_links = 96 + 12 * 4
        raw[root_links:root_links + 4] = (0x7FFFFFF0).to_bytes(4, "little")        # an inner link far out of range (fp32 record)
        q = 96 + n_nodes_f32 * 4 + 6 * 4
        raw[q:q + 4] = (0x7FFFFFF0).to_bytes(4, "little")                           # and in the fp16 record
        # recompute the file's checksum the way the builder does (FNV-1a over the four arrays)
        hsh = 1469598103934665603
        for byte in raw[96:]:
            hsh = ((hsh ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
        raw[88:96] = hsh.to_bytes(8, "little")
        with open(path, "wb") as f:
            f.write(raw)
        info6, h6, bad6, _ = build_host(sc.desc)
        assert info6.from_cache == 0 and bad6 == 0 and h6 == h_plain, "a structurally wrong cache file was trusted"

        def rewrite(mutate):
            """damage the (rebuilt, sound) file, then make its checksum right again: only the structural check can refuse it"""
            with open(path, "rb") as f:
                raw = bytearray(f.read())
            mutate(raw, int.from_bytes(raw[16:24], "little"), int.from_bytes(raw[24:32], "little"), int.from_bytes(raw[32:40], "little"))
            hsh = 1469598103934665603
            for byte in raw[96:]:
                hsh = ((hsh ^ byte) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
            raw[88:96] = hsh.to_bytes(8, "little")
            with open(path, "wb") as f:
                f.write(raw)
            info, h, bad, why = build_host(sc.desc)
            assert info.from_cache == 0 and bad == 0 and h == h_plain, "a damaged cache file with a valid checksum was trusted"
            assert build_host(sc.desc)[0].from_cache == 1      # ... and the rewritten file is sound again

        def bad_prim_index(raw, n_nodes_f32, n_nodes16, n_tris_f32):   # a leaf-order entry far outside the triangle array
            at = 96 + (n_nodes_f32 + n_nodes16 + n_tris_f32) * 4 + 4 * 1000
            raw[at:at + 4] = (0x7FFFFF00).to_bytes(4, "little")

        def zero_scale(raw, *_):                     # header scalar: frame_scale = 0 -> 1 / scale = inf in the kernels
            raw[60:64] = np.float32(0.0).tobytes()

        def nan_centre(raw, *_):
            raw[48:52] = np.float32(np.nan).tobytes()

        def odd_scale(raw, *_):                      # not a power of two
            raw[60:64] = (np.frombuffer(bytes(raw[60:64]), np.float32) * np.float32(1.5)).tobytes()

        def moved_triangle(raw, n_nodes_f32, n_nodes16, n_tris_f32):   # right id, wrong geometry: v0.x of record 5 nudged inside its leaf box
            at = 96 + (n_nodes_f32 + n_nodes16) * 4 + 5 * 48
            x = np.frombuffer(bytes(raw[at:at + 4]), np.float32)
            raw[at:at + 4] = np.nextafter(x, np.float32(np.inf)).astype(np.float32).tobytes()

        for m in (bad_prim_index, zero_scale, nan_centre, odd_scale, moved_triangle):
            rewrite(m)
    finally:
        lib.trh_set_bvh_cache_dir(None)
    assert build_host(sc.desc)[0].from_cache == 0
