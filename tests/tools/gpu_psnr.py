"""Development aid: PSNR / per-sample agreement of the HIP module vs the oracle per scene."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tracy_amd import Scene, HipTrace
from oracle.oracle import CpuRef

def srgb(x):
    x = np.clip(x, 0, 1); return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)
def psnr(a, b):
    m = np.mean((srgb(a.astype(np.float64)) - srgb(b.astype(np.float64))) ** 2); return 99 if m == 0 else 10 * np.log10(1 / m)

for name in ["cornell", "dragon", "spheres", "random", "testtree", "furnace", "helmet"]:
    for (w, h, spp) in [(64, 36, 16), (160, 90, 32), (160, 90, 128)]:
        sc = Scene(f"data/scenes/{name}.scn", w, h)
        ref = CpuRef(sc); mod = HipTrace(); mod.Startup(sc)
        ref.render(0, spp); mod.render(0, spp)
        img, _ = mod.resolve(); o = ref.output()
        d = np.abs(img[..., :3] - o).max(axis=-1)
        print(f"{name:9s} {w}x{h} {spp:4d}spp PSNR {psnr(img[...,:3], o):6.2f} dB  pixels differing>1e-4: {(d>1e-4).mean():.5f}  rays {ref.counters()['rays']} {mod.stats()['rays']}", flush=True)
        mod.Shutdown()
