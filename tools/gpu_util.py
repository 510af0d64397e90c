"""Development aid: per-phase lane utilisation from the instrumented kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace
for scene in sys.argv[1:] or ["dragon"]:
    sc = Scene(f"data/scenes/{scene}.scn", 1920, 1080)
    for pipe in (0, 2):
        mod = HipTrace(pipeline=pipe); mod.Startup(sc)
        mod.render(0, 64); mod.synchronize()   # the first launch records the tile costs, the second runs cost-ordered
        mod.stats(reset=True)
        mod.render(64, 64, stats=True)
        st = mod.stats(); d = mod.debug_counters()
        u_inner = (st["box_tests"] / 2) / (64 * max(1, d["wave_inner_iters"]))
        u_leaf = st["tri_tests"] / (64 * max(1, d["wave_leaf_iters"]))
        u_shade = st["rays"] / (64 * max(1, d["wave_shade_calls"]))
        u_gen = d["lanes_gen"] / (64 * max(1, d["wave_gen_calls"]))
        print(scene, "pipeline", pipe, f"lane utilisation: inner {u_inner:.3f} leaf {u_leaf:.3f} shade {u_shade:.3f} gen {u_gen:.3f}",
              f"| per ray: inner visits {st['box_tests']/2/st['rays']:.2f} tri {st['tri_tests']/st['rays']:.2f}",
              f"| wave iters per ray*64: inner {d['wave_inner_iters']*64/st['rays']:.2f} leaf {d['wave_leaf_iters']*64/st['rays']:.2f} shade {d['wave_shade_calls']*64/st['rays']:.2f}")
        if d["cycles_kernel"]:
            k = d["cycles_kernel"]
            if d.get("waves"):
                print(f"   load balance: mean wave lifetime / longest = {k / d['waves'] / max(1, d['cycles_longest_wave']):.3f} ({d['waves']} waves)")
            if d.get("waves") and d.get("t_last_end"):
                w, span = d["waves"], d["t_last_end"] - d["t_first_start"]
                print(f"   launch span {span} ticks; mean wave start {(d['t_sum_start'] / w - d['t_first_start']) / span:.4f} of the span after the first, "
                      f"mean wave end {(d['t_last_end'] - d['t_sum_end'] / w) / span:.4f} of the span before the last")
            if d.get("wave_inner_iters"):
                # the vote runs TRH_INNER_REPEAT (3) inner steps; the idle census is taken once per vote
                votes = max(1, d["idle_leaf_in_inner"] + d["idle_wait_in_inner"] + d["idle_none_in_inner"])
                per = lambda x: x / (64.0 * max(1, d["wave_inner_iters"] / 3.0))
                print(f"   lanes idle at an inner vote: on a leaf {per(d['idle_leaf_in_inner']):.3f}, waiting for shade {per(d['idle_wait_in_inner']):.3f}, no path {per(d['idle_none_in_inner']):.3f}")
            rest = k - d["cycles_inner"] - d["cycles_leaf"] - d["cycles_shade"] - d["cycles_refill"]
            print(f"   wave time split: inner {d['cycles_inner']/k:.3f} leaf {d['cycles_leaf']/k:.3f} shade {d['cycles_shade']/k:.3f} refill+generate {d['cycles_refill']/k:.3f} votes/tile bookkeeping {rest/k:.3f}")
        mod.Shutdown()
