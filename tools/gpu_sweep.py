"""Development aid: A/B timing of pipelines / knobs on the bench workload (one process per setting)."""
import os, sys, time, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from tracy_amd import Scene, HipTrace
    scene, w, h, spp, pipe = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    sc = Scene(f"data/scenes/{scene}.scn", w, h, subdivide=int(os.environ.get("TRH_SUBDIVIDE", "0")))
    mod = HipTrace(pipeline=pipe); mod.Startup(sc)
    mod.render(0, 4, stats=True); st0 = mod.stats(reset=True); mod.reset(); bi = mod.bvh_info()
    mod.render(0, spp); mod.synchronize(); mod.stats(reset=True)
    best = 1e9
    for i in range(3):
        t0 = time.time(); mod.render(1000 + i * spp, spp); mod.synchronize(); best = min(best, time.time() - t0)
    st = mod.stats(reset=True)
    print(json.dumps({"ms": best * 1e3, "msamples": w * h * spp / best / 1e6, "mrays": st["rays"] / 3 / best / 1e6,
                      "tris": bi["tri_count"], "nodes": bi["node_count"], "depth": bi["max_depth"], "build_ms": round(bi["build_ms"], 1), "builder": bi["builder"], "device_build_ms": round(bi["device_build_ms"], 2), "sah": round(bi["sah_cost"], 1),
                      "n_box": round(st0["box_tests"] / st0["rays"], 2), "n_tri": round(st0["tri_tests"] / st0["rays"], 2)}))
    sys.exit(0)
configs = []
for a in sys.argv[1:]:
    configs.append(a)
for cfg in configs or ["dragon:1920:1080:64"]:
    scene, w, h, spp = cfg.split(":")
    for pipe, thr, beta in [(2, 0, 4), (0, 24, 4), (1, 8, 4), (1, 16, 4), (1, 24, 4), (1, 32, 4)]:
        env = dict(os.environ); env["TRH_REFILL_THRESH"] = str(thr or 16); env["TRH_SCHED_BETA4"] = str(beta)
        out = subprocess.run([sys.executable, __file__, "child", scene, w, h, spp, str(pipe)], env=env, capture_output=True, text=True, timeout=300)
        print(cfg, "pipeline", pipe, "thresh", thr, "beta4", beta, out.stdout.strip() or out.stderr[-300:], flush=True)
