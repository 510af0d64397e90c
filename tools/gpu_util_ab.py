"""Development aid (GPU): phase times and lane utilisation of k_megakernel_ab from its instrumented build (TRH_STATS_AB=1).
   python tools/gpu_util_ab.py [scene ...] [--spp N] [--cfg cap,target,refill ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TRH_STATS_AB"] = "1"
from tracy_amd import Scene, HipTrace  # noqa: E402
args = sys.argv[1:]
spp = int(args[args.index("--spp") + 1]) if "--spp" in args else 64
cfgs = [a for a in args if "," in a] or ["1024,768,16"]
scenes = [a for a in args if a.isalpha()] or ["dragon"]
for scene in scenes:
    sc = Scene(f"data/scenes/{scene}.scn", 1920, 1080)
    for cfg in cfgs:
        cap, target, refill = cfg.split(",")
        os.environ.update(TRH_PQ_CAP=cap, TRH_PQ_TARGET=target, TRH_AB_REFILL=refill)
        mod = HipTrace(); mod.Startup(sc)
        mod.render(0, spp); mod.render(spp, spp); mod.synchronize(); mod.stats(reset=True)
        mod.render(2 * spp, spp); ms = mod.stats(reset=True)["kernel_ms"]
        mod.render(3 * spp, spp, stats=True)
        st = mod.stats(); d = mod.debug_counters()
        k = max(1, d["cycles_kernel"])
        ui = (st["box_tests"] / 2) / (64 * max(1, d["wave_inner_iters"]))
        ul = st["tri_tests"] / (64 * max(1, d["wave_leaf_iters"]))
        us = st["rays"] / (64 * max(1, d["wave_shade_calls"]))
        votes = 64.0 * max(1, d["wave_inner_iters"] / 3.0)
        rest = k - d["cycles_inner"] - d["cycles_leaf"] - d["cycles_shade"] - d["cycles_refill"]
        print(f"{scene} cap/target/refill {cfg}: {ms:.2f} ms | traverse-phase util inner {ui:.3f} leaf {ul:.3f}; shade util {us:.3f} | "
              f"idle at inner vote: leaf {d['idle_leaf_in_inner']/votes:.3f} finished {d['idle_wait_in_inner']/votes:.3f} no-ray {d['idle_none_in_inner']/votes:.3f} | "
              f"time: inner {d['cycles_inner']/k:.3f} leaf {d['cycles_leaf']/k:.3f} shade+push {d['cycles_shade']/k:.3f} camera {d['cycles_refill']/k:.3f} rest {rest/k:.3f} | "
              f"secondary rays: box/ray {st['box_tests']/max(1, st['rays'] - d['lanes_gen']):.1f} | Gcycles inner {d['cycles_inner']/1e9:.3f} leaf {d['cycles_leaf']/1e9:.3f} kernel {k/1e9:.3f}", flush=True)
        mod.Shutdown()
