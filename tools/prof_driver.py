"""Development aid (GPU): a few launches of one scene for rocprofv3 (tools/pmc_quick.sh).
   python3 tools/prof_driver.py [scene] [spp] [launches] [max_bounces] [pipeline]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "dragon"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
w, h = (3840, 2160) if name == "helmet" else (1920, 1080)
sc = Scene(f"data/scenes/{name}.scn", w, h)
if len(sys.argv) > 4 and int(sys.argv[4]) > 0:
    sc.desc.max_bounces = int(sys.argv[4])
mod = HipTrace(pipeline=int(sys.argv[5]) if len(sys.argv) > 5 else None)
mod.Startup(sc)
for k in range(n + 2):   # the first two launches (probe + first ordered launch) are warm-up
    mod.render(k * spp, spp)
mod.synchronize()
st = mod.stats()
print(f"{name} {spp} spp: {st['kernel_ms'] / st['launches']:.3f} ms per launch (all {st['launches']} launches)")
mod.Shutdown()
