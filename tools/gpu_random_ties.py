"""Development aid (GPU): classify every per-sample difference between the module and the oracle on random.scn."""
import os, sys, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, HipTrace
from oracle.oracle import CpuRef
name = sys.argv[1] if len(sys.argv) > 1 else "random"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
nspp = int(sys.argv[4]) if len(sys.argv) > 4 else 16
sc = Scene(f"data/scenes/{name}.scn", W, H)
mod = HipTrace(); mod.Startup(sc)
ref = CpuRef(sc)
classes = collections.Counter(); pairs = collections.Counter(); total = 0
for s in range(nspp):
    g = mod.sample_radiance(s); r = ref.sample_radiance(s)
    bad = (np.abs(g[..., :3] - r[..., :3]) > 1e-4 + 1e-4 * np.abs(r[..., :3])).any(axis=2) | (g[..., 3] != r[..., 3])
    ys, xs = np.nonzero(bad)
    total += len(ys)
    for y, x in zip(ys, xs):
        path = ref.trace_path(x, y, s)
        tie = [(hid, t, rid, rt, close) for hid, t, rid, rt, close in path if close > 0]
        if tie:
            hid, t, rid, rt, close = tie[0]
            same = (hid >> 24) == (rid >> 24)
            classes["tie, same mesh" if same else "tie, two meshes"] += 1
            pairs[(hid >> 24, rid >> 24, f"{abs(rt - t) / t:.1e}")] += 1
        else:
            classes["no rival within 1e-4 t"] += 1
            print("  unexplained:", s, x, y, g[y, x], r[y, x], path)
print(name, W, H, nspp, "spp: differing (pixel, sample) pairs:", total, "of", W * H * nspp, dict(classes))
for k, v in pairs.most_common(12):
    print("   mesh pair / dt/t:", k, v)
