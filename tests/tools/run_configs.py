"""Measure every BASELINE.json config on one GPU: throughput at the config's full sample count,
workload statistics from the instrumented kernel (n_box, n_tri, hit fraction, rays per
sample -> workload/<scene>.json), and parity against the oracle at the config's resolution
(primary-hit ids; PSNR of a 16-spp image after the build-owned sRGB tonemap).

    python tests/tools/run_configs.py [out.json]          (needs a GPU; the oracle legs run on the host cores)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tracy_amd import Scene, HipTrace  # noqa: E402
from oracle.oracle import CpuRef  # noqa: E402  (checker only)

CONFIGS = [
    # name, scene, W, H, total spp on one GPU, BASELINE.json config index
    ("plumbing", "random", 640, 360, 64, 0),
    ("cornell", "cornell", 1920, 1080, 1024, 1),
    ("spheres", "spheres", 1920, 1080, 1024, 2),
    ("dragon", "dragon", 1920, 1080, 4096, 3),
    ("helmet", "helmet", 3840, 2160, 1024, 4),   # 8192 spp over 8 GPUs = 1024 per GPU
]
PARITY_SPP = 16
SPP_PER_LAUNCH = 64


def srgb(x):
    x = np.clip(x, 0, 1)
    return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)


def psnr(a, b):
    m = np.mean((srgb(a.astype(np.float64)) - srgb(b.astype(np.float64))) ** 2)
    return 99.0 if m == 0 else float(10 * np.log10(1 / m))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "configs.json")
    only = set(sys.argv[2:])
    results = []
    for name, scene, w, h, spp, idx in CONFIGS:
        if only and name not in only:
            continue
        t0 = time.time()
        sc = Scene(f"data/scenes/{scene}.scn", w, h)
        mod = HipTrace()
        mod.Startup(sc)
        load_s = time.time() - t0
        # ---- workload statistics (instrumented kernel, 16 spp)
        mod.stats(reset=True)
        mod.render(0, PARITY_SPP, stats=True)
        st = mod.stats(reset=True)
        img16, _ = mod.resolve()
        wl = {
            "scene": scene, "width": w, "height": h, "spp": PARITY_SPP,
            "n_box": st["box_tests"] / st["rays"], "n_tri": st["tri_tests"] / st["rays"],
            "hit_frac": st["hits"] / st["rays"], "tex_fetches": st["tex_fetches"] / st["rays"],
            "rays_per_sample": st["rays"] / st["samples"],
        }
        os.makedirs(os.path.join(ROOT, "workload"), exist_ok=True)
        with open(os.path.join(ROOT, "workload", f"{scene}_{w}x{h}.json"), "w") as f:
            json.dump(wl, f, indent=1)
        # ---- throughput at the config's sample count
        mod.reset()
        mod.render(0, SPP_PER_LAUNCH)   # warm-up
        mod.synchronize()
        mod.reset()
        mod.stats(reset=True)
        t0 = time.time()
        done = 0
        while done < spp:
            n = min(SPP_PER_LAUNCH, spp - done)
            mod.render(done, n)
            done += n
        mod.synchronize()
        dt = time.time() - t0
        st = mod.stats(reset=True)
        info = mod.bvh_info()
        flops = st["rays"] * (25 * wl["n_box"] + 59 * wl["n_tri"] + 180 * wl["hit_frac"] + 12 * wl["tex_fetches"] + 25 * (1 - wl["hit_frac"]) + 8) + st["samples"] * 58
        # ---- parity at this resolution: primary-hit ids and a 16-spp image against the oracle
        ref = CpuRef(sc)
        t0 = time.time()
        ids_ref, t_ref, _ = ref.primary_hits()
        ids_gpu, t_gpu, _ = mod.primary_hits()
        ref.render(0, PARITY_SPP)
        oracle_s = time.time() - t0
        o = ref.output()
        same = float((ids_ref == ids_gpu).mean())
        r = {
            "config": name, "baseline_config_index": idx, "scene": scene, "width": w, "height": h, "spp": spp,
            "triangles": int(info["tri_count"]), "bvh_nodes": int(info["node_count"]), "bvh_depth": int(info["max_depth"]),
            "scene_load_and_upload_s": load_s, "render_s": dt,
            "msamples_per_s": w * h * spp / dt / 1e6, "mrays_per_s": st["rays"] / dt / 1e6,
            "kernel_ms_sum": st["kernel_ms"], "valu_tflops_algorithmic": flops / dt / 1e12, "valu_frac_of_157_3": flops / dt / 157.3e12,
            "workload": wl,
            "parity": {"primary_hit_ids_equal": same, "mismatching_pixels": int((ids_ref != ids_gpu).sum()),
                       "psnr_db_16spp": psnr(img16[..., :3], o), "oracle_seconds": oracle_s,
                       "oracle_msamples_per_s": w * h * PARITY_SPP / max(oracle_s, 1e-9) / 1e6},
        }
        print(json.dumps(r), flush=True)
        results.append(r)
        mod.Shutdown()
        del ref
    with open(out_path, "w") as f:
        json.dump(results, f, indent=1)


if __name__ == "__main__":
    main()
