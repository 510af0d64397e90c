#!/usr/bin/env python3
"""bench.py -- Msamples/s (+ Mrays/s) of the hip render module on the Stanford-dragon scene at
1920x1080 (BASELINE.json config 4), with inputs resident in HBM.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W      # one process per GPU, torch.distributed over RCCL
    python bench.py --gpus N --steps K --warmup W                   # ONE process, one host thread, N devices behind the C ABI
                                                                     # (trh_group_*: the shape Tracy's main() needs)

Default workload = BASELINE.json configs[3] as it is written: 4096 samples per pixel in all, sample-
partitioned over the N GPUs -- rank g owns the global sample indices [g*4096/N, (g+1)*4096/N)
(sharding.split_total_spp) -- and every rank traces its share in K equal steps (STRONG scaling:
the total work is fixed; 1 GPU, 20 steps: 204-205 spp per step). One step = one pass of the hot
path (trh_render) over the whole frame. --spp-per-step S switches to the weak-scaling form
(every rank traces S new samples per pixel per step, disjoint round-robin blocks).
The timed region holds exactly K steps on every rank plus the path's one exchange step (sum-
reduce of the float4 accumulation buffer to rank 0 over RCCL) and the device-side resolve;
no PCIe transfer is inside it. Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X vector FP32 peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0
RENDER_KERNEL = "k_megakernel_ab<false, true>"   # rocprofv3's name of the dominant kernel (default pipeline, fp16 node records)


def pmc_traffic(scene, W, H, spp):
    """HBM-side bytes of ONE launch of `spp` samples per pixel from the PMC counters (profiles/pmc_traffic.json, written by
    tools/profile_bench.sh from separate --pmc passes: 2 x FETCH_SIZE + WRITE_SIZE as the MICROARCH guide prescribes). An
    exact entry for this launch size if the profile has one, else the profile's linear model bytes = fixed + per_spp * spp
    (measured at two or more launch sizes of this workload), else None."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        t = json.load(open(path))
    except Exception:
        return None
    key = f"{scene}_{W}x{H}"
    if f"{key}_{spp}spp" in t:
        return t[f"{key}_{spp}spp"]
    m = t.get(f"{key}_model")
    if m:
        return m["fixed_bytes"] + m["bytes_per_spp"] * spp
    return None



def algorithmic_flops(st):
    """F_ray = 25*n_box + 59*n_tri + 180*[hit] + 12*n_tex + 25*[miss] + 8*[RR],
    + 58 per sample (camera ray 50 + accumulate 8). n_box / n_tri / hits / tex are COUNTED by the
    instrumented kernel on this very workload (trh_render_stats), not assumed."""
    miss = st["rays"] - st["hits"]
    return (25.0 * st["box_tests"] + 59.0 * st["tri_tests"] + 180.0 * st["hits"] + 12.0 * st["tex_fetches"]
            + 25.0 * miss + 8.0 * st["hits"] + 58.0 * st["samples"])


def algorithmic_bytes(st, node_record_bytes):
    """Secondary figure: one inner-node record per 2 box tests (32 B fp16 records, 64 B in the fp32
    fallback), 48 B per triangle test, 172 B per hit (64 B shading record + hit bookkeeping), 16 B
    per texel, 32 B per sample."""
    return (0.5 * node_record_bytes * st["box_tests"] + 48.0 * st["tri_tests"] + 172.0 * st["hits"] + 16.0 * st["tex_fetches"]
            + 32.0 * st["samples"])


def cpu_baseline(scene_name, width, height, budget_s=12.0):
    """The CPU oracle (oracle/cpu_ref.cpp, a restatement of the reference's OpenMP module) timed on
    this host's cores on a bounded sample of the same workload: whole frames (1 spp each) of the
    same scene at the same resolution until ~budget_s of wall time is used."""
    from tracy_amd.scene import Scene
    from oracle.oracle import CpuRef, lib
    sc = Scene(f"data/scenes/{scene_name}.scn", width, height)
    ref = CpuRef(sc)
    threads = lib().oref_num_threads()
    t0 = time.time()
    ref.render(0, 1)
    first = time.time() - t0
    n = int(max(1, min(256, budget_s / max(first, 1e-3))))
    ref.counters(reset=True)
    t0 = time.time()
    ref.render(1, n)
    dt = time.time() - t0
    rays = ref.counters()["rays"]
    return {"value": width * height * n / dt / 1e6, "unit": "Msamples/s", "mrays_per_s": rays / dt / 1e6, "cores": threads,
            "kind": "port", "sample": f"{scene_name} {width}x{height}, {n} spp (whole frames), reference kd-tree, OpenMP {threads} threads, {dt:.1f} s"}


def psnr_vs_oracle(mod, scene, spp):
    """PSNR of the module's image against the CPU oracle's at equal spp (same sample set), both sRGB-encoded and
    clamped to [0, 1] -- the north_star's correctness figure (b), at a sample count the oracle finishes in seconds."""
    from oracle.oracle import CpuRef

    def srgb(x):
        x = np.clip(x.astype(np.float64), 0.0, 1.0)
        return np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(x, 1 / 2.4) - 0.055)

    mod.bind_accum_buffer(None, 0)
    mod.reset()
    mod.render(0, spp)
    img, _ = mod.resolve()
    ref = CpuRef(scene)
    ref.render(0, spp)
    mse = float(np.mean((srgb(img[..., :3]) - srgb(ref.output())) ** 2))
    return 200.0 if mse == 0.0 else 10.0 * np.log10(1.0 / mse)


def main_group(args):
    """--gpus N without a launcher: ONE process, one host thread, N devices behind the C ABI (trh_group_*,
    csrc/hip/tracy_group.hip) -- the shape the reference's single-threaded main() needs (src/win_raytracer.cpp:480,494-517).
    Every timed step is one trh_group_render of total_spp / K samples, split over the devices by sample index
    ([g*S/G, (g+1)*S/G) of the step's S samples: N launches in flight from this thread); the timed region ends with
    trh_group_resolve: the one ncclReduce(sum, f32, W*H*4, root 0) inside ncclGroupStart/End + the mean on device 0."""
    import ctypes as C
    from tracy_amd import Scene, HipTrace, HipTraceGroup, capi
    n = C.c_int(0)
    capi.hip_lib().trh_device_count(C.byref(n))
    rehearsal = n.value < args.gpus
    if rehearsal:
        if not args.alias_devices:
            raise SystemExit(f"--gpus {args.gpus} but this box has {n.value} HIP device(s); --alias-devices rehearses the N>1 control flow on shared devices")
        os.environ["TRH_GROUP_ALIAS_DEVICE"] = "1"
    scene = Scene(f"data/scenes/{args.scene}.scn", args.width, args.height)
    W, H = scene.width, scene.height
    if args.total_spp < args.steps * args.gpus:
        raise SystemExit(f"--total-spp {args.total_spp} over {args.gpus} devices and {args.steps} steps leaves launches without samples")
    S = -(-args.total_spp // args.steps)                    # samples per step over all devices (ceil)
    S_dev = -(-S // args.gpus)                              # ... and per device launch

    def step_range(k):
        if k < args.warmup:
            return args.total_spp + k * S, S
        j = k - args.warmup
        lo, hi = (j * args.total_spp) // args.steps, ((j + 1) * args.total_spp) // args.steps
        return lo, hi - lo

    # ---- instrumented pass (untimed, one device): the roofline numerator for one device launch of this workload
    one = HipTrace(device=0)
    one.Startup(scene)
    one.stats(reset=True)
    one.render(0, S_dev, stats=True)
    st_algo = one.stats(reset=True)
    bvh = one.bvh_info()
    one.Shutdown()

    grp = HipTraceGroup(ngpus=args.gpus)
    grp.Startup(scene)
    if grp.ngpus != args.gpus:
        raise SystemExit(f"the group has {grp.ngpus} members, not {args.gpus}")
    step = 0
    for _ in range(args.warmup):
        grp.render(*step_range(step))
        step += 1
    grp.resolve(to_host=False)         # warms the exchange step up too (communicator, reduced buffer)
    grp.reset()
    grp.synchronize()
    grp.stats(reset=True)
    first_timed = step_range(step)[0]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        grp.render(*step_range(step))
        step += 1
    grp.resolve(to_host=False)         # exchange step + mean on device 0; returns when every device's stream has drained
    dt = time.perf_counter() - t0
    st = grp.stats(reset=False)
    t1 = time.perf_counter()
    img, spp_done = grp.resolve(to_host=True)
    d2h_ms = (time.perf_counter() - t1) * 1e3
    if args.save:
        np.save(args.save, img)
    kernel_ms = st["kernel_ms"] / max(1, st["launches"])    # slowest device: mean over its launches
    launch_scale = (args.total_spp / args.steps / args.gpus) / S_dev
    flops_launch = algorithmic_flops(st_algo) * launch_scale
    achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
    out = {
        "metric": "Msamples/s @1080p dragon scene (path tracing, <=5 bounces, Russian roulette)",
        "value": st["samples"] / dt / 1e6, "unit": "Msamples/s", "mrays_per_s": st["rays"] / dt / 1e6,
        "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"{args.scene}.scn {W}x{H}, {scene.tri_count} triangles, max 5 bounces + Russian roulette, seed 0xABCDEF "
                                f"(BASELINE.json configs[3]); {args.total_spp} spp in all, {args.steps} steps of {args.total_spp // args.steps}-{S} spp, "
                                f"every step sample-partitioned over {args.gpus} device(s) ([g*S/G,(g+1)*S/G))"),
                   "spp_per_step": S, "spp_per_device_launch": S_dev, "total_spp": int(spp_done), "first_sample_index": first_timed,
                   "parallelism": (f"ONE process, one host thread: trh_group_* over {args.gpus} devices, one ncclReduce(sum) to device 0 per resolve"
                                   + (" (ALIAS REHEARSAL on fewer devices: members share GPUs, device-side sum -- not a measurement)" if rehearsal else "")),
                   "pipeline": "megakernel", "bvh": bvh, "rays_per_sample": st["rays"] / max(1, st["samples"])},
        "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_TFLOPS,
                     "traffic": pmc_traffic(args.scene, W, H, S_dev), "kernel": RENDER_KERNEL, "kernel_ms_avg": kernel_ms,
                     "spp_per_launch": S_dev, "flops_per_launch": flops_launch, "note": "per device (slowest member)"},
        "d2h_resolve_ms": d2h_ms,
    }
    print(json.dumps(out), flush=True)
    grp.Shutdown()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--total-spp", type=int, default=4096, help="samples per pixel of the whole job (BASELINE.json configs[3]), split over ranks and steps")
    ap.add_argument("--spp-per-step", type=int, default=None, help="override: weak scaling, every rank traces this many samples per step")
    ap.add_argument("--psnr-spp", type=int, default=16, help="rank 0, N=1, outside the timed region: PSNR against the CPU oracle at this many spp (0 = skip)")
    ap.add_argument("--scene", default="dragon")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pipeline", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--save", default=None, help="rank 0: write the resolved image as .npy")
    ap.add_argument("--alias-devices", action="store_true",
                    help="single-process form on a box with fewer GPUs than --gpus: the group's members share devices "
                         "(TRH_GROUP_ALIAS_DEVICE=1, device-side sum instead of RCCL). A REHEARSAL of the N>1 control flow, not a measurement")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (the product path). gloo = rehearsal of the N>1 control flow on a box with fewer GPUs "
                         "than ranks: ranks share devices and the exchange step is staged through host memory")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from tracy_amd import Scene, HipTrace
    from tracy_amd import sharding

    rank, world, local_rank = sharding.env_rank_world()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU fallback)")
    if world == 1 and args.gpus > 1:
        return main_group(args)      # no launcher: this one process drives the N devices itself (trh_group_*)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}, "
                         f"or without a launcher (one process drives the devices through trh_group_*)")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)

    scene = Scene(f"data/scenes/{args.scene}.scn", args.width, args.height)
    mod = HipTrace(device=local_rank, pipeline=args.pipeline)
    mod.Startup(scene)
    W, H = scene.width, scene.height
    weak = args.spp_per_step is not None
    if weak:
        S = args.spp_per_step                      # samples per launch of the instrumented / roofline pass

        def step_range(k):                         # k counts warm-up and timed steps alike
            return sharding.step_sample_range(rank, world, k, S)
    else:
        my_begin, my_count = sharding.split_total_spp(rank, world, args.total_spp)
        if my_count < args.steps:
            raise SystemExit(f"--total-spp {args.total_spp} over {world} ranks and {args.steps} steps leaves steps without samples")
        S = -(-my_count // args.steps)             # the larger of the two step sizes (ceil)

        def step_range(k):
            # warm-up steps trace samples beyond the job's range (same launch shape); timed step j = k - warmup
            # owns [j*my_count/K, (j+1)*my_count/K) of this rank's share
            if k < args.warmup:
                return args.total_spp + (k * world + rank) * S, S
            j = k - args.warmup
            lo, hi = (j * my_count) // args.steps, ((j + 1) * my_count) // args.steps
            return my_begin + lo, hi - lo

    # accumulation buffer owned by torch so that torch.distributed (RCCL) can reduce it in place
    accum = torch.zeros((H, W, 4), dtype=torch.float32, device=f"cuda:{local_rank}")
    mod.bind_accum_buffer(accum.data_ptr(), accum.numel() * 4)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- instrumented pass (untimed): the roofline numerator for one launch of this workload
    mod.stats(reset=True)
    mod.render(step_range(args.warmup)[0], S, stats=True)
    st_algo = mod.stats(reset=True)
    accum.zero_()
    torch.cuda.synchronize()   # zero_() runs on torch's stream, the module renders on its own

    # ---- warmup
    step = 0
    for _ in range(args.warmup):
        mod.render(*step_range(step))
        step += 1
    mod.synchronize()
    if world > 1:
        sharding.reduce_accum_to_root(accum.clone(), dist)   # RCCL warm-up on a scratch copy
    accum.zero_()
    torch.cuda.synchronize()
    mod.stats(reset=True)
    first_timed = step_range(step)[0]

    # ---- timed region: exactly K steps + the exchange step + device-side resolve
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        mod.render(*step_range(step))
        step += 1
    mod.synchronize()
    sharding.reduce_accum_to_root(accum, dist if world > 1 else None)
    if world > 1:
        torch.cuda.synchronize()     # the reduce runs on torch's RCCL stream, the resolve on the module's
    if rank == 0:
        mod.resolve(to_host=False)
    barrier()
    dt = time.perf_counter() - t0

    st = mod.stats(reset=False)
    t_all = torch.tensor([dt, float(st["rays"]), float(st["samples"])], dtype=torch.float64,
                         device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
    if world > 1:
        tmax = t_all.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t_all.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt = float(tmax[0])
        total_rays, total_samples = float(tsum[1]), float(tsum[2])
    else:
        total_rays, total_samples = float(st["rays"]), float(st["samples"])

    if rank == 0:
        # PCIe-inclusive figure, reported separately (never `value`)
        t1 = time.perf_counter()
        img, _ = mod.resolve(to_host=True)
        d2h_ms = (time.perf_counter() - t1) * 1e3
        if args.save:
            np.save(args.save, img)
        kernel_ms = st["kernel_ms"] / max(1, st["launches"])
        # the instrumented launch traced S spp; the timed launches average my_count / K spp (204.8 at the defaults)
        launch_scale = 1.0 if weak else (my_count / args.steps) / S
        flops_launch = algorithmic_flops(st_algo) * launch_scale
        bvh = mod.bvh_info()
        bytes_launch = algorithmic_bytes(st_algo, bvh["node_bytes"] / max(1, bvh["node_count"])) * launch_scale
        achieved = flops_launch / (kernel_ms * 1e-3) / 1e12
        traffic = pmc_traffic(args.scene, W, H, S)
        out = {
            "metric": "Msamples/s @1080p dragon scene (path tracing, <=5 bounces, Russian roulette)",
            "value": total_samples / dt / 1e6,
            "unit": "Msamples/s",
            "mrays_per_s": total_rays / dt / 1e6,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": (f"{args.scene}.scn {W}x{H}, {scene.tri_count} triangles, max 5 bounces + Russian roulette, seed 0xABCDEF "
                                    f"(BASELINE.json configs[3]); " +
                                    (f"weak form: {S} spp per step per GPU" if weak else
                                     f"{args.total_spp} spp in all, sample-partitioned over {world} GPU(s) ([g*S/G,(g+1)*S/G)), "
                                     f"each share traced in {args.steps} steps of {my_count // args.steps}-{S} spp")),
                       "spp_per_step": S, "total_spp": int(round(total_samples / (W * H))), "first_sample_index": first_timed,
                       "parallelism": f"sample-index sharding x{world}, one sum-reduce to rank 0" + ("" if args.backend == "nccl" else " (gloo REHEARSAL, not a measurement)"),
                       "pipeline": {0: "megakernel (camera / traverse / shade phases per wave)", 1: "wavefront", 2: "megakernel_simple", 3: "megakernel_lanes (round 2)"}[args.pipeline or 0],
                       "bvh": bvh, "rays_per_sample": total_rays / total_samples},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic,
                         "kernel": RENDER_KERNEL if not args.pipeline else f"pipeline {args.pipeline}", "kernel_ms_avg": kernel_ms,
                         "spp_per_launch": S,
                         "flops_per_launch": flops_launch,
                         "per_ray": {"box_tests": st_algo["box_tests"] / st_algo["rays"], "tri_tests": st_algo["tri_tests"] / st_algo["rays"],
                                     "hit_frac": st_algo["hits"] / st_algo["rays"], "tex_fetches": st_algo["tex_fetches"] / st_algo["rays"]},
                         # secondary: the algorithmic bytes are served by L1/L2/Infinity Cache (hot data = 7.5 MB);
                         # `traffic` above is what actually reaches the fabric. Not an HBM-bound path.
                         "bytes_secondary": {"algorithmic_bytes_per_launch": bytes_launch,
                                             "algorithmic_GBs_all_cache_levels": bytes_launch / (kernel_ms * 1e-3) / 1e9,
                                             "hbm_peak_GBs": PEAK_HBM_GBS,
                                             "hbm_frac_from_pmc_traffic": (traffic / (kernel_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if traffic else None}},
            "d2h_resolve_ms": d2h_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.scene, W, H)
            if args.psnr_spp > 0:
                out["psnr_vs_cpu_ref_db"] = psnr_vs_oracle(mod, scene, args.psnr_spp)
                out["psnr_note"] = (f"sRGB-encoded, clamped to [0,1], peak 1; GPU module vs CPU oracle, both at samples [0,{args.psnr_spp}) "
                                    f"of the same workload, outside the timed region")
        print(json.dumps(out), flush=True)

    mod.bind_accum_buffer(None, 0)
    mod.Shutdown()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
