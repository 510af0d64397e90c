"""Multi-GPU behind the C ABI (trh_group_*, include/tracy_hip.h): one host thread drives N devices, the samples of a
render call are split by sample index ([g*S/N, (g+1)*S/N) on device g, BASELINE.json configs[3]) and one RCCL reduce
to device 0 produces the image. Reference caller: src/win_raytracer.cpp:480,511-517 (single device there)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import load_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _differing(a, b):
    """number of 32-bit words whose bits differ"""
    return int((np.asarray(a).view(np.uint32) != np.asarray(b).view(np.uint32)).sum())


def test_split_samples_partitions_every_call():
    from tracy_amd import split_samples
    for n in (1, 2, 3, 4, 8):
        for begin, count in ((0, 4096), (17, 1), (5, 7), (100, 8), (0, 0), (4294967000, 295)):
            ranges = [split_samples(begin, count, n, g) for g in range(n)]
            pos = begin
            for b, c in ranges:          # consecutive, disjoint, in device order
                assert b == pos
                pos += c
            assert pos == begin + count
            if count:
                assert max(c for _, c in ranges) - min(c for _, c in ranges) <= 1
    # the split BASELINE.json names: 4096 spp over 8 GPUs = 512 each, GPU g starts at g * 512
    assert [split_samples(0, 4096, 8, g) for g in range(8)] == [(g * 512, 512) for g in range(8)]


def test_group_entry_points_fail_loudly_without_a_group():
    from tracy_amd import capi
    lib = capi.hip_lib()
    assert lib.trh_group_render(None, 0, 1, 0) != 0
    assert b"NULL" in lib.trh_last_error()
    assert lib.trh_group_resolve(None, None, None) != 0
    assert lib.trh_group_size(None) == 0


@pytest.mark.gpu
def test_one_device_group_is_the_plain_context_bit_for_bit(gpu_available):
    assert gpu_available, "needs a HIP device"
    from tracy_amd import HipTrace, HipTraceGroup
    sc = load_scene("dragon", 480, 270)
    one = HipTrace(device=0)
    one.Startup(sc)
    one.render(0, 32)
    one.render(32, 5)
    ref, n_ref = one.resolve()
    pic_ref = one.present()
    rays_ref = one.stats()["rays"]
    one.Shutdown()

    grp = HipTraceGroup(ngpus=1)
    grp.Startup(sc)
    assert grp.ngpus == 1
    grp.render(0, 32)
    grp.render(32, 5)
    img, n = grp.resolve()
    pic = grp.present()
    rays = grp.stats()["rays"]
    grp.Shutdown()
    assert n == n_ref == 37
    assert _differing(img, ref) == 0
    assert _differing(pic, pic_ref) == 0
    assert rays == rays_ref


_FORCED = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from tracy_amd import Scene, HipTraceGroup
sc = Scene("data/scenes/cornell.scn", 320, 180)
g = HipTraceGroup(ngpus=1)
g.Startup(sc)
g.render(0, 16)
img, n = g.resolve()
img2, _ = g.resolve()          # the per-device sums survive a resolve
g.render(16, 16)
img3, n3 = g.resolve()
pic = g.present()
np.savez({out!r}, img=img, img2=img2, img3=img3, pic=pic, n=n, n3=n3)
g.Shutdown()
print("ok")
"""


@pytest.mark.gpu
def test_rccl_reduce_path_on_one_device(gpu_available, tmp_path):
    """TRH_GROUP_FORCE_RCCL=1: a one-device group goes through dlopen(librccl), ncclCommInitAll, ncclReduce inside
    ncclGroupStart/End on the render stream, and resolves from the reduced buffer. In a child process: RCCL is only
    ever loaded there."""
    assert gpu_available, "needs a HIP device"
    out = str(tmp_path / "forced.npz")
    env = dict(os.environ, TRH_GROUP_FORCE_RCCL="1")
    r = subprocess.run([sys.executable, "-c", _FORCED.format(root=ROOT, out=out)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    z = np.load(out)
    from tracy_amd import HipTrace
    sc = load_scene("cornell", 320, 180)
    one = HipTrace(device=0)
    one.Startup(sc)
    one.render(0, 16)
    ref, _ = one.resolve()
    one.render(16, 16)
    ref3, _ = one.resolve()
    pic = one.present()
    one.Shutdown()
    assert int(z["n"]) == 16 and int(z["n3"]) == 32
    assert _differing(z["img"], ref) == 0     # a 1-rank sum is a copy
    assert _differing(z["img2"], ref) == 0
    assert _differing(z["img3"], ref3) == 0
    assert _differing(z["pic"], pic) == 0


@pytest.mark.gpu
def test_two_device_group_equals_one_device(gpu_available):
    assert gpu_available, "needs a HIP device"
    import ctypes as C
    from tracy_amd import capi, HipTrace, HipTraceGroup
    n = C.c_int(0)
    capi.hip_lib().trh_device_count(C.byref(n))
    if n.value < 2:
        pytest.skip("one HIP device on this box: the 2-device group needs two")
    sc = load_scene("dragon", 480, 270)
    one = HipTrace(device=0)
    one.Startup(sc)
    one.render(0, 64)
    ref, _ = one.resolve()
    rays_ref = one.stats()["rays"]
    one.Shutdown()
    grp = HipTraceGroup(ngpus=2)
    grp.Startup(sc)
    assert grp.ngpus == 2
    grp.render(0, 64)
    img, ns = grp.resolve()
    rays = grp.stats()["rays"]
    grp.Shutdown()
    assert ns == 64
    assert rays == rays_ref                      # the same sample set, whatever the device count
    np.testing.assert_allclose(img[..., :3], ref[..., :3], rtol=2e-6, atol=1e-7)   # float summation order only
    assert _differing(np.ascontiguousarray(img[..., 3]), np.ascontiguousarray(ref[..., 3])) == 0


# ---- the N > 1 path on a one-GPU box: TRH_GROUP_ALIAS_DEVICE=1 lets the members of a group share device 0 (own context,
# stream, scene replica and accumulation buffer each); the exchange step is a device-side sum where RCCL would refuse
# duplicate devices. Everything else is the code a real N-GPU group runs (tracy_group.hip).

@pytest.mark.gpu
@pytest.mark.parametrize("members", [2, 3])
def test_alias_group_equals_one_context(gpu_available, monkeypatch, members):
    assert gpu_available, "needs a HIP device"
    import ctypes as C
    from tracy_amd import capi, HipTrace, HipTraceGroup
    sc = load_scene("dragon", 480, 270)
    one = HipTrace(device=0)
    one.Startup(sc)
    one.render(0, 32)
    ref32, _ = one.resolve()
    one.render(32, 32)
    ref, _ = one.resolve()
    pic_ref = one.present()
    rays_ref = one.stats()["rays"]
    bvh_ref = one.bvh_info()
    one.Shutdown()

    monkeypatch.setenv("TRH_GROUP_ALIAS_DEVICE", "1")
    grp = HipTraceGroup(ngpus=members)
    grp.Startup(sc)
    assert grp.ngpus == members
    # the host-built BVH is built once and uploaded to every member
    lib = capi.hip_lib()
    for i in range(members):
        ctx = C.c_void_p()
        assert lib.trh_group_context(grp._grp, i, C.byref(ctx)) == 0
        b = capi.trh_bvh_info()
        assert lib.trh_get_bvh_info(ctx, C.byref(b)) == 0
        assert (b.node_count, b.tri_count, b.max_depth) == (bvh_ref["node_count"], bvh_ref["tri_count"], bvh_ref["max_depth"])
    grp.render(0, 32)
    mid, n_mid = grp.resolve()       # a resolve in the middle leaves the per-member sums intact
    mid2, _ = grp.resolve()
    grp.render(32, 32)
    img, ns = grp.resolve()
    pic = grp.present()
    rays = grp.stats()["rays"]
    grp.Shutdown()
    assert n_mid == 32 and ns == 64
    assert rays == rays_ref                      # the same sample set, whatever the member count
    assert _differing(mid, mid2) == 0
    for got, want in ((mid, ref32), (img, ref)):
        np.testing.assert_allclose(got[..., :3], want[..., :3], rtol=2e-6, atol=1e-7)   # float summation order only
        assert _differing(np.ascontiguousarray(got[..., 3]), np.ascontiguousarray(want[..., 3])) == 0
    assert (pic != pic_ref).mean() < 1e-3        # 8-bit picture: a rounding flip here and there at most


@pytest.mark.gpu
def test_alias_group_render_failure_leaves_a_consistent_state(gpu_available, monkeypatch):
    """A member that fails its launch after earlier members were enqueued (injected: TRH_GROUP_FAIL_MEMBER) must not
    leave accum_spp out of step with the buffers unnoticed: the group refuses render / resolve / present until
    trh_group_reset_accumulation, and is then as good as new. A failure of member 0 (nothing launched yet) changes nothing."""
    assert gpu_available, "needs a HIP device"
    from tracy_amd import HipError, HipTraceGroup
    sc = load_scene("cornell", 320, 180)
    monkeypatch.setenv("TRH_GROUP_ALIAS_DEVICE", "1")
    grp = HipTraceGroup(ngpus=3)
    grp.Startup(sc)
    grp.render(0, 12)
    good, n = grp.resolve()
    assert n == 12
    monkeypatch.setenv("TRH_GROUP_FAIL_MEMBER", "0")
    with pytest.raises(HipError, match="injected"):
        grp.render(12, 12)
    same, n = grp.resolve()                      # nothing was launched: state unchanged
    assert n == 12 and _differing(same, good) == 0
    monkeypatch.setenv("TRH_GROUP_FAIL_MEMBER", "2")
    with pytest.raises(HipError, match="injected"):
        grp.render(12, 12)
    monkeypatch.delenv("TRH_GROUP_FAIL_MEMBER")
    for call in (lambda: grp.render(24, 3), grp.resolve, grp.present):
        with pytest.raises(HipError, match="part-way"):
            call()
    grp.reset()
    grp.render(0, 12)
    again, n = grp.resolve()
    grp.Shutdown()
    assert n == 12 and _differing(again, good) == 0
