"""Development aid: time alternative builds of libtracy_hip.so (tools/variants/*.so) on the bench workload."""
import os, sys, subprocess, shutil, glob
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(ROOT, "tracy_amd", "lib", "libtracy_hip.so")
shutil.copy(lib, "/tmp/_orig.so")
args = sys.argv[1:] or ["dragon", "1920", "1080", "64", "0"]
try:
    for v in ["/tmp/_orig.so"] + sorted(glob.glob(os.path.join(ROOT, "tools", "variants", "*.so"))):
        shutil.copy(v, lib)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_sweep.py"), "child", *args], capture_output=True, text=True, timeout=300)
        print(os.path.basename(v), out.stdout.strip() or out.stderr[-300:], flush=True)
finally:
    shutil.copy("/tmp/_orig.so", lib)
