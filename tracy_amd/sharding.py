"""Sample-index sharding across GPUs (one process per GPU, torch.distributed; the backend
"nccl" is RCCL on ROCm, "gloo" in the CPU tests).

The path shards by SAMPLE INDEX: every rank renders all pixels for a
disjoint range of global sample indices; the RNG state of a path depends only on
(pixel, global sample index, seed), so the union over ranks is exactly the sample set one GPU
would render. Each rank accumulates a full-frame float4 buffer (rgb sums, .w = count); the only
exchange step is ONE sum-reduce of that buffer to rank 0, followed by mean = sum / count there.
The reference has no multi-device path; this is new functionality behind the same module API.
"""
import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def step_sample_range(rank, world, step, spp_per_step):
    """Weak scaling: every step, every rank renders spp_per_step new samples per pixel.
    Global sample indices are dealt round-robin in blocks: block (step * world + rank)."""
    begin = (step * world + rank) * spp_per_step
    return begin, spp_per_step


def split_total_spp(rank, world, total_spp):
    """Strong scaling (north_star wording): GPU g owns [g*S/G, (g+1)*S/G)."""
    begin = (rank * total_spp) // world
    end = ((rank + 1) * total_spp) // world
    return begin, end - begin


def reduce_accum_to_root(accum, dist=None):
    """The one exchange step: in-place sum-reduce of the accumulation buffer to rank 0."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return accum
    if accum.is_cuda and dist.get_backend() == "gloo":
        # rehearsal / CPU-test backend: stage through host memory (RCCL reduces device memory in place)
        host = accum.cpu()
        dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
        accum.copy_(host)
        return accum
    dist.reduce(accum, dst=0, op=dist.ReduceOp.SUM)
    return accum


def resolve_mean(accum):
    """mean = rgb sum / count (host/torch version of k_resolve, for the CPU tests)."""
    w = accum[..., 3:4]
    return accum[..., :3] / w.clamp(min=1.0)
