"""Development aid (GPU): the shader clock the megakernel really runs at = sum over waves of s_memtime ticks / sum over
waves of s_memrealtime ticks (100 MHz), from the instrumented kernel's per-wave stamps (trh_get_debug_counters)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace  # noqa: E402

for name in sys.argv[1:] or ["dragon", "spheres", "cornell"]:
    sc = Scene(f"data/scenes/{name}.scn", 1920, 1080)
    mod = HipTrace()
    mod.Startup(sc)
    for k in range(20):                     # warm: ~0.25 s of back-to-back launches before the stamped one
        mod.render(k * 64, 64)
    mod.synchronize()
    mod.stats(reset=True)
    mod.render(2000, 64, stats=True)
    st = mod.stats()
    d = mod.debug_counters()
    real = (d["t_sum_end"] - d["t_sum_start"]) / 100e6      # wave-seconds
    clock = d["cycles_kernel"] / real / 1e9
    span_ms = (d["t_last_end"] - d["t_first_start"]) / 100e6 * 1e3
    print(f"{name:8s}: instrumented launch {st['kernel_ms']:.2f} ms (span of the waves {span_ms:.2f} ms), shader clock inside the kernel {clock:.3f} GHz", flush=True)
    mod.Shutdown()
