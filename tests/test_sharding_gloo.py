"""N > 1 host path on CPU: world_size-2 `gloo` run of the sample-index sharding + the one
sum-reduce to rank 0 + mean resolve (tracy_amd/sharding.py, the same code bench.py drives over
RCCL). The per-rank renderer here is the CPU oracle (allowed in tests), so the test checks the
sharding contract itself: the union of the ranks' sample sets equals the single-rank sample set
and the reduced image equals the single-rank image."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, total_spp, mode, out_path):
    sys.path.insert(0, ROOT)
    os.environ["OMP_NUM_THREADS"] = "2"
    from tracy_amd.scene import Scene
    from tracy_amd import sharding
    from oracle.oracle import CpuRef
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    sc = Scene("data/scenes/cornell.scn", 48, 27)
    ref = CpuRef(sc)
    accum = torch.zeros((27, 48, 4), dtype=torch.float32)
    if mode == "strong":
        ranges = [sharding.split_total_spp(rank, world, total_spp)]
    else:
        steps = total_spp // (world * 2)
        ranges = [sharding.step_sample_range(rank, world, k, 2) for k in range(steps)]
    mine = []
    for begin, count in ranges:
        for s in range(begin, begin + count):
            r = torch.from_numpy(ref.sample_radiance(s))
            accum[..., :3] += r[..., :3]
            accum[..., 3] += 1.0
            mine.append(s)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    sharding.reduce_accum_to_root(accum, dist)
    if rank == 0:
        img = sharding.resolve_mean(accum)
        np.savez(out_path, img=img.numpy(), count=accum[..., 3].numpy(), samples=np.array(sorted(sum(gathered, []))))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["weak", "strong"])
def test_two_rank_sharding_equals_single_rank(tmp_path, mode, scene_loader, oracle_mod):
    total = 8
    out = str(tmp_path / f"{mode}.npz")
    mp.spawn(_worker, args=(2, _free_port(), total, mode, out), nprocs=2, join=True)
    got = np.load(out)
    assert got["samples"].tolist() == list(range(total))          # disjoint and complete
    assert (got["count"] == total).all()
    ref = oracle_mod.CpuRef(scene_loader("cornell", 48, 27))
    acc = np.zeros((27, 48, 3), dtype=np.float64)
    for s in range(total):
        acc += ref.sample_radiance(s)[..., :3]
    assert np.allclose(got["img"], acc / total, rtol=2e-6, atol=1e-7)


def test_ranges_are_disjoint_and_cover():
    from tracy_amd import sharding
    for world in (1, 2, 3, 4, 8):
        seen = []
        for step in range(5):
            for r in range(world):
                b, n = sharding.step_sample_range(r, world, step, 16)
                seen += list(range(b, b + n))
        assert sorted(seen) == list(range(5 * world * 16))
        seen = []
        for r in range(world):
            b, n = sharding.split_total_spp(r, world, 4096 + 3)
            seen += list(range(b, b + n))
        assert seen == list(range(4096 + 3))
