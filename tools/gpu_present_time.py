"""Development aid: cost of the display-side calls (present / resolve to host) next to a 1-spp update."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace
sc = Scene("data/scenes/dragon.scn", 1920, 1080)
mod = HipTrace(); mod.Startup(sc)
mod.render(0, 8); mod.synchronize(); mod.present(); mod.resolve()
def t(f, n=20):
    t0 = time.time()
    for _ in range(n): f()
    return (time.time() - t0) / n * 1e3
k = [8]
def upd():
    mod.render(k[0], 1); mod.synchronize(); k[0] += 1
print("OnUpdate 1 spp %.3f ms | present (tonemap + 8 MB to host) %.3f ms | resolve to host (33 MB) %.3f ms | resolve on device %.3f ms" %
      (t(upd), t(mod.present), t(lambda: mod.resolve(to_host=True)), t(lambda: mod.resolve(to_host=False))))
