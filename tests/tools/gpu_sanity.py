"""Quick GPU sanity run (development aid): parity vs the oracle at small size + a timing at 1080p."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tracy_amd import Scene, HipTrace
from oracle.oracle import CpuRef

for name in sys.argv[1:] or ["cornell", "dragon", "spheres", "random", "testtree", "furnace"]:
    sc = Scene(f"data/scenes/{name}.scn", 320, 180)
    ref = CpuRef(sc)
    mod = HipTrace()
    mod.Startup(sc)
    print(name, sc.tri_count, mod.bvh_info(), flush=True)
    ids_r, t_r, uv_r = ref.primary_hits()
    ids_g, t_g, uv_g = mod.primary_hits()
    mism = ids_r != ids_g
    print("  primary hits: mismatches", int(mism.sum()), "of", ids_r.size, "t maxabs (matching)", float(np.abs(t_r - t_g)[~mism & (ids_r != 0xffffffff)].max(initial=0)))
    for s in (0, 1):
        a = ref.sample_radiance(s); b = mod.sample_radiance(s)
        close = np.isclose(a[..., :3], b[..., :3], rtol=1e-4, atol=1e-5).all(axis=-1)
        print(f"  sample {s}: pixels equal(1e-4) {close.mean():.5f}  bounce-count equal {(a[...,3]==b[...,3]).mean():.5f} mean ref {a[...,:3].mean():.5f} gpu {b[...,:3].mean():.5f}")
    ref.render(0, 8)
    mod.render(0, 8)
    img, n = mod.resolve()
    o = ref.output()
    print("  8spp: max abs diff", float(np.abs(img[..., :3] - o).max()), "mean", float(o.mean()), float(img[..., :3].mean()), "rays", ref.counters()["rays"], mod.stats()["rays"])
    mod.Shutdown()

sc = Scene("data/scenes/dragon.scn", 1920, 1080)
mod = HipTrace(); mod.Startup(sc)
mod.render(0, 1); mod.synchronize(); mod.stats(reset=True)
for spp in (4, 16, 64):
    t0 = time.time(); mod.render(100, spp); mod.synchronize(); dt = time.time() - t0
    st = mod.stats(reset=True)
    print(f"dragon 1080p {spp} spp: {dt*1e3:.1f} ms wall, kernel {st['kernel_ms']:.1f} ms, {st['samples']/dt/1e6:.1f} Msamples/s {st['rays']/dt/1e6:.1f} Mrays/s", flush=True)
mod.render(0, 4, stats=True); st = mod.stats(reset=True)
print("stats", st)
