"""Parity stress: random cameras (inside and outside the geometry, arbitrary orientation) on every scene;
primary-hit ids, t, uv against the oracle, plus per-sample radiance of one sample index. Round 3: the DEFAULT pipeline
(camera packets + per-wave rings) renders 4 spp from every camera and must equal the sum of the four per-sample images of
the probe kernel (one lane, one path, its own stack walk); and a second context on the device-built tree must give the
host-built tree's primary hits.

    python tests/tools/fuzz_cameras.py [cameras_per_scene]      (needs a GPU)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tracy_amd import Scene, HipTrace, capi  # noqa: E402
from oracle.oracle import CpuRef  # noqa: E402

MISS = 0xFFFFFFFF
n_cams = int(sys.argv[1]) if len(sys.argv) > 1 else 25
rng = np.random.default_rng(2024)
tot_px = tot_bad = tot_img = tot_dev = 0
for name in ["cornell", "dragon", "spheres", "random", "testtree", "furnace", "helmet", "bunny", "trimesh"]:
    sc = Scene(f"data/scenes/{name}.scn", 160, 90)
    v = sc.vertices()[:, :3]
    lo, hi = v.min(axis=0), v.max(axis=0)
    ctr, ext = (lo + hi) / 2, float(np.max(hi - lo))
    mod = HipTrace(); mod.Startup(sc)
    dev = HipTrace(builder=capi.TRH_BUILDER_DEVICE_LBVH); dev.Startup(sc)
    ref = CpuRef(sc)
    bad_scene = rad_bad = img_bad = dev_bad = 0
    for k in range(n_cams):
        r = ext * (0.05 + 1.5 * rng.random())                       # from deep inside to well outside
        eye = ctr + r * rng.normal(size=3) / np.sqrt(3)
        target = ctr + 0.3 * ext * (rng.random(3) - 0.5)
        up = rng.normal(size=3)
        sc.set_view(eye.astype(np.float32), target.astype(np.float32), up.astype(np.float32))
        e, ivp = sc.eye_ivp()
        mod.set_camera(e, ivp); ref.set_camera(e, ivp)
        ids, t, uv = mod.primary_hits()
        rids, rt, ruv = ref.primary_hits()
        bad = ids != rids
        same = ~bad & (ids != MISS)
        assert np.array_equal(t[same], rt[same]) and np.array_equal(uv[same], ruv[same]), f"{name} cam {k}: t/u/v differ on equal ids"
        bad_scene += int(bad.sum())
        a, b = mod.sample_radiance(3), ref.sample_radiance(3)
        ok = np.isclose(a[..., :3], b[..., :3], rtol=1e-4, atol=1e-5).all(axis=-1) & (a[..., 3] == b[..., 3])
        rad_bad += int((~ok).sum())
        # the default pipeline against the probe kernel, sample for sample
        mod.reset(); mod.render(0, 4)
        img, _ = mod.resolve()
        want = np.zeros_like(a[..., :3], dtype=np.float64)
        for smp in range(4):
            want += mod.sample_radiance(smp)[..., :3]
        okimg = np.isclose(img[..., :3], want / 4, rtol=2e-5, atol=1e-6).all(axis=-1)
        img_bad += int((~okimg).sum())
        # the device-built tree against the host-built one
        dev.set_camera(e, ivp)
        dids, dt, duv = dev.primary_hits()
        dev_bad += int((dids != ids).sum())
    n = n_cams * 160 * 90
    tot_px += n; tot_bad += bad_scene
    tot_img += img_bad; tot_dev += dev_bad
    print(f"{name:9s} {n_cams} cameras: hit-id mismatches {bad_scene} of {n} ({bad_scene/n:.2e}), sample-radiance outliers {rad_bad} ({rad_bad/n:.2e}), "
          f"default pipeline vs probe kernel (4 spp): {img_bad} pixels differ, device-built tree vs host-built: {dev_bad} ids differ (builder {dev.bvh_info()['builder']})", flush=True)
    mod.Shutdown(); dev.Shutdown()
print(f"TOTAL hit-id mismatches {tot_bad} of {tot_px} ({tot_bad/tot_px:.2e}); default pipeline vs probe kernel: {tot_img} pixels; device-built vs host-built tree: {tot_dev} ids")
