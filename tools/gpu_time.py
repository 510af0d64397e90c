"""Development aid (GPU): HIP-event time per launch of the megakernel on the bench scenes, for A/B runs of kernel builds
(TRACY_HIP_LIBRARY selects the library).   python tools/gpu_time.py [scene ...] [--spp N]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
spp = int(sys.argv[sys.argv.index("--spp") + 1]) if "--spp" in sys.argv else 64
args = [a for a in args if not a.isdigit()]
for name in args or ["dragon", "spheres", "cornell", "helmet"]:
    w, h = (3840, 2160) if name == "helmet" else (1920, 1080)
    sc = Scene(f"data/scenes/{name}.scn", w, h)
    mod = HipTrace()
    mod.Startup(sc)
    mod.render(0, spp)
    mod.render(spp, spp)
    mod.synchronize()
    best = []
    for rep in range(3):
        mod.stats(reset=True)
        for k in range(6):
            mod.render((2 + rep * 6 + k) * spp, spp)
        st = mod.stats(reset=True)
        best.append(st["kernel_ms"] / st["launches"])
    ms = min(best)
    print(f"{name:8s} {w}x{h} {spp} spp/launch: {ms:7.3f} ms (runs {' '.join('%.3f' % b for b in best)}), {w * h * spp / ms / 1e3:8.0f} Msamples/s", flush=True)
    mod.Shutdown()
