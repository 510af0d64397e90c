import sys, time
sys.path.insert(0, '.')
from tracy_amd import Scene, HipTrace
sc = Scene("data/scenes/dragon.scn", 1920, 1080)
warm = HipTrace(); warm.Startup(sc); warm.render(0, 64); warm.synchronize(); warm.Shutdown()   # load code objects
for spp in (64, 256):
    mod = HipTrace(); mod.Startup(sc)
    ts = []
    for k in range(3):
        t0 = time.time(); mod.render(k * spp, spp); mod.synchronize(); ts.append((time.time() - t0) * 1e3)
    print(spp, "spp: first launch %.2f ms, second %.2f, third %.2f" % tuple(ts), flush=True)
    mod.Shutdown()
