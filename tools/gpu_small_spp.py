"""Development aid: kernel time vs wall time of launches with few samples per pixel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace
scene = sys.argv[1] if len(sys.argv) > 1 else "dragon"
sc = Scene(f"data/scenes/{scene}.scn", 1920, 1080)
mod = HipTrace(); mod.Startup(sc)
for spp in (1, 2, 4, 8, 16, 64):
    mod.render(0, spp); mod.synchronize(); mod.stats(reset=True)
    n = 50
    t0 = time.time()
    for i in range(n):
        mod.render(100 + i * spp, spp); mod.synchronize()
    wall = (time.time() - t0) / n * 1e3
    st = mod.stats(reset=True)
    print(f"spp {spp:3d}: wall {wall:7.3f} ms/update, events {st['kernel_ms']/st['launches']:7.3f} ms/update, {1920*1080*spp/wall/1e3:8.1f} Msamples/s", flush=True)
mod.Shutdown()
