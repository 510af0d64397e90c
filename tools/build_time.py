"""Development aid: host BVH build time vs builder threads (no GPU needed)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tracy_amd import Scene, capi
lib = capi.hip_lib()
for lv in (0, 1, 2):
    sc = Scene("data/scenes/dragon.scn", 64, 36, subdivide=lv)
    for t in ("1", "4", "16"):
        os.environ["TRH_BUILD_THREADS"] = t
        best = 1e9
        for _ in range(2):
            info, h, bad = capi.trh_bvh_info(), C.c_uint64(), C.c_uint64()
            lib.trh_bvh_build_host(C.byref(sc.desc), C.byref(info), C.byref(h), C.byref(bad))
            best = min(best, info.build_ms)
        print(f"{sc.tri_count:8d} triangles, {t:>2s} threads: {best:7.1f} ms  (violations {bad.value}, hash {h.value:016x})", flush=True)
