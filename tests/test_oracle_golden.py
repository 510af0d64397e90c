"""CPU: the oracle against the committed golden fixtures (tests/golden/make_golden.py) and
against known answers derivable from the scene files themselves."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

SCENES = ["cornell", "dragon", "spheres", "random", "testtree", "furnace", "helmet", "bunny", "trimesh"]


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_golden(oracle_mod, scene_loader, name):
    g = np.load(os.path.join(GOLDEN, f"{name}_64x36.npz"))
    ref = oracle_mod.CpuRef(scene_loader(name, 64, 36))
    ids, t, uv = ref.primary_hits()
    assert np.array_equal(ids, g["ids"])                      # integer: bit exact
    assert np.array_equal(t, g["t"]) and np.array_equal(uv, g["uv"])
    # same binary on the same ISA level reproduces the floats exactly; libm differences between
    # hosts could move sin/cos/atan2 by an ulp, hence a tolerance instead of array_equal
    s0 = ref.sample_radiance(0)
    assert np.allclose(s0, g["s0"], rtol=1e-5, atol=1e-6)
    ref.render(0, 16)
    assert np.allclose(ref.output(), g["mean16"], rtol=1e-4, atol=1e-5)
    assert ref.counters()["rays"] == int(g["rays16"])


def test_render_is_deterministic_and_thread_count_independent(oracle_mod, scene_loader):
    sc = scene_loader("cornell", 64, 36)
    a = oracle_mod.CpuRef(sc)
    b = oracle_mod.CpuRef(sc)
    a.render(0, 4)
    b.render(0, 2)
    b.render(2, 2)
    assert np.array_equal(a.output(), b.output())
    assert a.counters()["rays"] == b.counters()["rays"]


def test_running_mean_equals_mean_of_samples(oracle_mod, scene_loader):
    # CPUDetails::UpdateOutput (cpu_details.cpp:187-199) is a running mean of per-sample radiance
    ref = oracle_mod.CpuRef(scene_loader("dragon", 64, 36))
    samples = np.stack([ref.sample_radiance(s)[..., :3] for s in range(6)])
    ref.render(0, 6)
    assert np.allclose(ref.output(), samples.astype(np.float64).mean(axis=0), rtol=1e-5, atol=1e-6)
    # ray counter = sum of bounce-loop iterations (cpu_trace.cpp:115-116)
    iters = sum(int(ref.sample_radiance(s)[..., 3].sum()) for s in range(6))
    assert ref.counters()["rays"] == iters


def test_camera_cut_resets_blend_counter(oracle_mod, scene_loader):
    ref = oracle_mod.CpuRef(scene_loader("cornell", 64, 36))
    ref.render(0, 3)
    ref.reset()                      # eCameraCut -> ResetFrameCounter (cpu_trace.cpp:76-78)
    ref.render(0, 1)                 # blend factor 0: the old image is overwritten
    assert np.array_equal(ref.output(), ref.sample_radiance(0)[..., :3])


def test_testtree_hit_map_known_answer(oracle_mod, scene_loader):
    # 4 single-triangle objects (testtree.scn:56-59); packed id = mesh << 24 | 0
    g = np.load(os.path.join(GOLDEN, "testtree_64x36.npz"))
    ids = g["ids"]
    present = set(np.unique(ids).tolist())
    assert present <= {0x00000000, 0x01000000, 0x02000000, 0x03000000, 0xFFFFFFFF}
    assert 0x03000000 in present      # the big yellow floor triangle is visible
    # camera above the plane y=-2 looking down: the hit points of mesh 3 lie in that plane
    sc = scene_loader("testtree", 64, 36)
    ref = oracle_mod.CpuRef(sc)
    _, t, _ = ref.primary_hits()
    assert (t[ids == 0xFFFFFFFF] == np.float32(3.402823466e38)).all()


def test_furnace_sphere_radiance(oracle_mod, scene_loader):
    """furnace.scn: white sky, grey 0.18 sphere, "sphere color should be exactly 0.18"
    (furnace.scn:2-3). With this BRDF (material.h:250-261) 10 % + Fresnel of the samples take
    the 0.85-white specular branch, so the converged value is above 0.18; recorded, not asserted
    to be 0.18. What must hold exactly: every path ends on the sky, radiance = product of the
    attenuations, each either 0.18 or 0.85."""
    ref = oracle_mod.CpuRef(scene_loader("furnace", 64, 36))
    ids, _, _ = ref.primary_hits()
    acc = np.zeros((36, 64, 3), dtype=np.float64)
    n = 64
    for s in range(n):
        r = ref.sample_radiance(s)
        acc += r[..., :3]
    sphere = (ids != 0xFFFFFFFF)
    inner = sphere & np.roll(sphere, 1, 0) & np.roll(sphere, -1, 0) & np.roll(sphere, 1, 1) & np.roll(sphere, -1, 1)
    val = (acc / n)[inner].mean()
    assert 0.18 < val < 0.40
    assert np.allclose((acc / n)[~sphere & ~np.roll(sphere, 1, 0) & ~np.roll(sphere, -1, 0) & ~np.roll(sphere, 1, 1) & ~np.roll(sphere, -1, 1)], 1.0, atol=1e-6)


def test_serial_rng_mode_matches_reference_stream_usage(oracle_mod, scene_loader):
    # the literal reference behaviour: one stream for the whole frame, x-major pixel order
    sc = scene_loader("testtree", 64, 36)
    a = oracle_mod.CpuRef(sc)
    a.render_serial_rng(1)
    out = a.output()
    assert np.isfinite(out).all() and out.max() <= 1.0 + 1e-6
    b = oracle_mod.CpuRef(sc)
    b.render_serial_rng(1)
    assert np.array_equal(out, b.output())
