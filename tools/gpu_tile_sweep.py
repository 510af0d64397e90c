"""Development aid (GPU): time per launch against the tile size (TRH_TILE_PLOG2, read once per process: one child per value).
   python tools/gpu_tile_sweep.py [scene] [--spp a,b,c] [--plog2 a,b,c]"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    from tracy_amd import Scene, HipTrace
    scene, spps = sys.argv[2], [int(x) for x in sys.argv[3].split(",")]
    sc = Scene(f"data/scenes/{scene}.scn", 1920, 1080)
    mod = HipTrace(); mod.Startup(sc)
    out = []
    for spp in spps:
        for k in range(3): mod.render(k * spp, spp)
        mod.synchronize(); mod.stats(reset=True)
        n = max(4, min(40, 2000 // spp))
        for i in range(n): mod.render(1000 + i * spp, spp)
        st = mod.stats(reset=True)
        out.append(f"{spp}: {st['kernel_ms'] / st['launches']:.3f}")
    print("  ".join(out), flush=True)
    mod.Shutdown()
    sys.exit(0)
args = sys.argv[1:]
scene = next((a for a in args if a.isalpha()), "dragon")
spps = args[args.index("--spp") + 1] if "--spp" in args else "16,26,51,102,205"
plog2 = args[args.index("--plog2") + 1].split(",") if "--plog2" in args else ["-1", "3", "4", "5", "6", "7", "8"]
for p in plog2:
    env = dict(os.environ)
    if p != "-1": env["TRH_TILE_PLOG2"] = p
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child", scene, spps], env=env, capture_output=True, text=True, timeout=300)
    print(f"{scene} plog2 {p:>2s} | ms per launch at spp  {r.stdout.strip() or r.stderr[-400:]}", flush=True)
    if r.returncode != 0: sys.exit(1)
