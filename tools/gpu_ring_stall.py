import os, sys
sys.path.insert(0, os.getcwd())
os.environ["TRH_STATS_AB"] = "1"
from tracy_amd import Scene, HipTrace
for scene in ("dragon", "cornell"):
    sc = Scene(f"data/scenes/{scene}.scn", 1920, 1080)
    mod = HipTrace(); mod.Startup(sc)
    mod.render(0, 205); mod.render(205, 205); mod.synchronize(); mod.stats(reset=True)
    mod.render(410, 205, stats=True)
    st = mod.stats(); d = mod.debug_counters()
    k = d["cycles_kernel"]
    print(scene, "wave cycles total", k, "shade-batch ring stall", d["idle_none_in_inner"] * 16 / k, "refill ring stall", d["idle_wait_in_inner"] * 16 / k,
          "phases inner/leaf/shade/camera", d["cycles_inner"] / k, d["cycles_leaf"] / k, d["cycles_shade"] / k, d["cycles_refill"] / k)
    mod.Shutdown()
