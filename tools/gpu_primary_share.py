"""Development aid: how much of a launch is primary rays (max_bounces = 1) and how well do they use the lanes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace
for scene, w, h in (("dragon", 1920, 1080), ("cornell", 1920, 1080), ("spheres", 1920, 1080)):
    for mb in (5, 1):
        sc = Scene(f"data/scenes/{scene}.scn", w, h)
        sc.desc.max_bounces = mb
        mod = HipTrace(); mod.Startup(sc)
        mod.render(0, 64); mod.synchronize(); mod.stats(reset=True)
        best = 1e9
        for i in range(3):
            t0 = time.time(); mod.render(1000 + 64 * i, 64); mod.synchronize(); best = min(best, time.time() - t0)
        mod.stats(reset=True)
        mod.render(5000, 16, stats=True)
        st = mod.stats(); d = mod.debug_counters()
        print(f"{scene} max_bounces {mb}: {best*1e3:6.2f} ms per 64 spp, rays/sample {st['rays']/st['samples']:.2f}, "
              f"inner util {(st['box_tests']/2)/(64*max(1,d['wave_inner_iters'])):.2f} leaf util {st['tri_tests']/(64*max(1,d['wave_leaf_iters'])):.2f} "
              f"shade util {st['rays']/(64*max(1,d['wave_shade_calls'])):.2f}, box/ray {st['box_tests']/st['rays']:.1f} tri/ray {st['tri_tests']/st['rays']:.2f}", flush=True)
        mod.Shutdown()
