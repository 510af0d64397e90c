"""Generates tests/golden/*.npz from the CPU oracle (oracle/cpu_ref.cpp).

    python tests/golden/make_golden.py

The reference holds no golden vectors for this path and cannot be built here (SURVEY.md
section 8c), so these fixtures pin the ORACLE'S OWN outputs ("parity unpinned" w.r.t. a
reference binary). They travel to the GPU box, where the HIP module is compared to them
without needing the oracle build to behave identically on another host CPU.
Inputs: assets/data/scenes/<scene>.scn at 64x36, seed 0xABCDEF.
Outputs per scene: primary-hit packed ids, t, (u,v); radiance of sample 0 and sample 5 with
bounce counts; 16-spp running-mean image; ray count of those 16 frames.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tracy_amd.scene import Scene  # noqa: E402
from oracle.oracle import CpuRef  # noqa: E402

W, H, SPP = 64, 36, 16
SCENES = ["cornell", "dragon", "spheres", "random", "testtree", "furnace", "helmet", "bunny", "trimesh"]

if __name__ == "__main__":
    for name in SCENES:
        sc = Scene(f"data/scenes/{name}.scn", W, H)
        ref = CpuRef(sc)
        ids, t, uv = ref.primary_hits()
        s0 = ref.sample_radiance(0)
        s5 = ref.sample_radiance(5)
        ref.render(0, SPP)
        out = ref.output()
        rays = ref.counters()["rays"]
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"{name}_64x36.npz"), ids=ids, t=t, uv=uv, s0=s0, s5=s5,
                            mean16=out, rays16=np.uint64(rays))
        print(name, "hit fraction %.3f" % (ids != 0xFFFFFFFF).mean(), "mean16", out.mean(), "rays", rays)
