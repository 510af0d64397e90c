"""Parity at the FULL frame size of every GPU workload BASELINE.json names (configs[1..4]): the HIP module through the
C ABI against the CPU oracle over the whole frame.

  (a) primary-ray hit ids: >= 99.99 % of the pixels equal, t/u/v bit-equal wherever the ids are, every mismatch
      classified as a tie between two hits at the same distance (collision.h:66: last tested wins; the BVH has no
      such order) -- never a hole;
  (b) the 16-spp image: PSNR >= 40 dB after the same sRGB encoding (peak 1), ray counts within 0.5 %;
  plus the north_star target itself at a sample count the oracle traces in seconds: dragon 1920x1080, 64 spp.
The oracle renders the same sample set (RNG keyed by pixel, sample index, seed), so the differences are float
ordering (libm vs ocml transcendentals) and traversal ties only.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MISS = 0xFFFFFFFF

# scene, width, height  (BASELINE.json configs[1], [2], [3] at 1080p; configs[4] at 4K)
CONFIGS = [("cornell", 1920, 1080), ("spheres", 1920, 1080), ("dragon", 1920, 1080), ("helmet", 3840, 2160)]


def srgb(x):
    x = np.clip(x.astype(np.float64), 0.0, 1.0)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)


def psnr(a, b):
    mse = float(np.mean((srgb(a) - srgb(b)) ** 2))
    return 200.0 if mse == 0.0 else 10.0 * np.log10(1.0 / mse)


@pytest.mark.parametrize("name,w,h", CONFIGS)
def test_full_frame_hit_ids_and_16spp_image_vs_oracle(oracle_mod, name, w, h):
    from tracy_amd import Scene, HipTrace
    sc = Scene(f"data/scenes/{name}.scn", w, h)     # not cached: a 4K helmet scene holds 320 MiB of texels
    ref = oracle_mod.CpuRef(sc)
    mod = HipTrace()
    mod.Startup(sc)
    try:
        ids_r, t_r, uv_r = ref.primary_hits()
        ids_g, t_g, uv_g = mod.primary_hits()
        mism = ids_r != ids_g
        assert mism.mean() <= 1e-4, f"{name}: {int(mism.sum())} hit-id mismatches of {mism.size}"
        ok = ~mism & (ids_r != MISS)
        assert int((t_r[ok].view(np.uint32) != t_g[ok].view(np.uint32)).sum()) == 0, "t differs on pixels with equal ids"
        assert int((uv_r[ok].view(np.uint32) != uv_g[ok].view(np.uint32)).sum()) == 0, "u,v differ on pixels with equal ids"
        for y, x in zip(*np.nonzero(mism)):
            assert ids_r[y, x] != MISS and ids_g[y, x] != MISS, f"{name} ({x},{y}): one side missed where the other hit"
            assert abs(float(t_r[y, x]) - float(t_g[y, x])) <= 4e-6 * abs(float(t_r[y, x])) + 1e-6, f"{name} ({x},{y}): not a tie"
        spp = 16
        mod.render(0, spp)
        img, n = mod.resolve()
        ref.render(0, spp)
        out = ref.output()
        assert n == spp and int((img[..., 3] != spp).sum()) == 0
        p = psnr(img[..., :3], out)
        assert p >= 40.0, f"{name} {w}x{h}: PSNR {p:.1f} dB at {spp} spp"
        r_ref, r_gpu = ref.counters()["rays"], mod.stats()["rays"]
        assert abs(r_ref - r_gpu) <= 0.005 * r_ref, f"{name}: rays {r_gpu} vs oracle {r_ref}"
        print(f"{name} {w}x{h}: ids equal {1.0 - mism.mean():.7f} ({int(mism.sum())} ties), PSNR@{spp}spp {p:.1f} dB, rays {r_gpu}/{r_ref}")
    finally:
        mod.Shutdown()
        ref.close()


def test_dragon_1080p_64spp_psnr_target(oracle_mod):
    """north_star: 'renders the Stanford dragon scene at 1920x1080 with PSNR >= 40 dB against the CPU reference at
    equal spp' -- here at 64 spp, traced as the bench traces them (one launch of 64 spp)."""
    from tracy_amd import Scene, HipTrace
    sc = Scene("data/scenes/dragon.scn", 1920, 1080)
    ref = oracle_mod.CpuRef(sc)
    mod = HipTrace()
    mod.Startup(sc)
    try:
        mod.render(0, 64)
        img, n = mod.resolve()
        ref.render(0, 64)
        p = psnr(img[..., :3], ref.output())
        assert n == 64 and p >= 40.0, f"PSNR {p:.1f} dB"
        r_ref, r_gpu = ref.counters()["rays"], mod.stats()["rays"]
        assert abs(r_ref - r_gpu) <= 0.002 * r_ref
        print(f"dragon 1920x1080 64 spp: PSNR {p:.1f} dB, rays {r_gpu}/{r_ref}")
    finally:
        mod.Shutdown()
        ref.close()


def test_random_scn_every_difference_is_classified(oracle_mod, scene_loader):
    """random.scn is the weakest parity number (45-56 dB where the other scenes reach 67 dB and more). Every (pixel, sample)
    whose radiance or bounce count differs from the oracle's at 1920x1080 is traced again in the oracle bounce by bounce
    (oref_trace_path: hit, t, and the nearest rival candidate from a brute-force pass over all triangles) and put in a class:
      * tie: some bounce of the path has a rival of ANOTHER mesh within 1e-5 t of the hit -- the light box (mesh 2) whose
        face lies IN the wall's face (mesh 10, z = 1.6). The reference keeps whichever it tests last and prunes the later
        mesh by the strict slab test `tmin < minT` (collision.h:66,130; cpu_details.cpp:128-142): with t_b == t_a to the last
        bit that is a comparison of two roundings of the same distance. accept_candidate replays the rule, and on camera
        rays (bit-equal on both sides) the replay is exact -- asserted below: no tie difference at bounce 0. On later bounces
        the module's ray differs from the oracle's in the last bits (ocml against libm sinf / cosf in Scatter), so the coin
        lands differently now and then: not a traversal property, and no wider tie window changes it (dt/t <= 2e-7 here);
      * no rival: an ulp-level flip of a hit / miss or roulette decision somewhere along the path.
    Bars: tie class <= 5e-5 of all (pixel, sample) pairs, no-rival class <= 5e-6, nothing at bounce 0."""
    import collections
    sc = scene_loader("random", 1920, 1080)
    from tracy_amd import HipTrace
    mod = HipTrace(device=0)
    mod.Startup(sc)
    ref = oracle_mod.CpuRef(sc)
    nspp = 4
    classes, tie_bounce, pairs = collections.Counter(), collections.Counter(), collections.Counter()
    for s in range(nspp):
        g, r = mod.sample_radiance(s), ref.sample_radiance(s)
        bad = (np.abs(g[..., :3] - r[..., :3]) > 1e-4 + 1e-4 * np.abs(r[..., :3])).any(axis=2) | (g[..., 3] != r[..., 3])
        for y, x in zip(*np.nonzero(bad)):
            path = ref.trace_path(x, y, s)
            ties = [(b, hid, rid, abs(rt - t) / t) for b, (hid, t, rid, rt, close) in enumerate(path) if close > 0 and (hid >> 24) != (rid >> 24) and abs(rt - t) <= 1e-5 * t]
            if ties:
                b, hid, rid, rel = ties[0]
                classes["tie"] += 1
                tie_bounce[b] += 1
                pairs[tuple(sorted((hid >> 24, rid >> 24)))] += 1
                assert rel <= 1e-6
            else:
                classes["no rival"] += 1
    mod.Shutdown()
    total = 1920 * 1080 * nspp
    print(f"random.scn 1920x1080, {nspp} spp: {dict(classes)} of {total} (pixel, sample) pairs; tie bounce {dict(tie_bounce)}; mesh pairs {dict(pairs)}")
    assert classes["tie"] <= 5e-5 * total
    assert classes["no rival"] <= 5e-6 * total
    assert tie_bounce[0] == 0, "a tie decided differently on a camera ray: the replay of the reference's rule is broken"
    assert set(pairs) <= {(2, 10)}, "ties between meshes other than the light box and the wall it lies in"
