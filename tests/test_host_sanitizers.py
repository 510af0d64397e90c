"""Host code under AddressSanitizer + UBSan (g++, CPU only; the GPU pool runs no sanitizers): the
texture decoders and the .scn / OBJ readers are fed deterministic mutations of the shipped assets
(truncation, byte noise, damaged headers and tables, marker runs). They may refuse a file; they must
not crash, overflow, or read/write out of bounds."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "tracy_amd", "csrc", "host")


@pytest.fixture(scope="module")
def fuzz_binary(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("san") / "decoder_fuzz")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-ffp-contract=off",
           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "decoder_fuzz.cpp"),
           os.path.join(HOST, "image_decode.cpp"), os.path.join(HOST, "scene_host.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


def write_hdr(path):
    rng = np.random.default_rng(3)
    w, h = 64, 24
    px = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n" + f"-Y {h} +X {w}\n".encode())
        for y in range(h):
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                row = px[y, :, c]
                for i in range(0, w, 16):
                    f.write(bytes([16]) + row[i:i + 16].tobytes())


@pytest.mark.parametrize("scene", ["trimesh"])
def test_damaged_files_under_asan_ubsan(fuzz_binary, tmp_path, scene):
    hdr = str(tmp_path / "probe.hdr")
    write_hdr(hdr)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_binary, os.path.join(ROOT, "assets", "data", "earth.jpg"), hdr, "60",
                        os.path.join(ROOT, "assets", "data", "scenes", f"{scene}.scn"), os.path.join(ROOT, "assets"), str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "decoded" in r.stdout
    counts = [int(x) for x in r.stdout.replace(",", " ").split() if x.isdigit()]
    assert counts[0] > 0 and counts[1] > 0 and counts[2] > 0      # some files survive, some are refused, some scenes load
    assert counts[3] > 0 and counts[4] >= 8                       # OBJ: well-formed ones parse, every bad-index case is refused


@pytest.mark.parametrize("flags", [["-fsanitize=thread"], ["-fsanitize=address,undefined", "-fno-sanitize-recover=all"]], ids=["tsan", "asan_ubsan"])
def test_parallel_bvh_builder_under_sanitizers(tmp_path, flags):
    """Data races in the task-parallel builder would show up as TSan reports; the soup sizes cover the
    empty scene, the single-leaf root, the parallel sub-tree tasks and the multi-thread binning of big nodes."""
    out = str(tmp_path / "bvh_check")
    cmd = ["g++", "-std=c++17", "-O1", "-g", *flags, "-pthread", os.path.join(ROOT, "tests", "cpp", "bvh_builder_check.cpp"),
           os.path.join(HOST, "bvh_builder.cpp"), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "bvh builder ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
