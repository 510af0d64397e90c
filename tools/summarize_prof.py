"""Summarises a tools/profile_bench.sh output directory: kernel stats, per-kernel PMC averages of the timed launches, the three
KPIs the round-2 verdict asked for, and the fabric-traffic figures bench.py reads (profiles/pmc_traffic.json)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
RENDER = "k_megakernel_ab<false"
print(f"# rocprofv3 summary ({os.path.basename(d)}): `bench.py --gpus 1 --steps 20 --warmup 5` (the driver's command)\n")
avg_ms = None
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel-trace --stats\n")
    print("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.3f} | {r['Percentage']} |")
        if RENDER in r["Name"]:
            avg_ms = float(r["AverageNs"]) / 1e6
    print()


def collect(pattern):
    agg = defaultdict(lambda: defaultdict(list))
    for f in glob.glob(os.path.join(d, pattern, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return agg


def launches_of(cs, spp_launches):
    """the LAST spp_launches dispatches of the render kernel = the timed steps (earlier ones: tile-order probe launches of 1 spp
    and the warm-up steps)"""
    out = {}
    for c, v in cs.items():
        vals = [x[1] for x in sorted(v)][-spp_launches:]
        out[c] = sum(vals) / len(vals)
    return out


agg = collect("pmc*")
print("## PMC (average over the 20 timed launches of 204-205 spp; one rocprofv3 --pmc run per counter group)\n")
out = {}
for k, cs in agg.items():
    if "megakernel" not in k:
        continue
    timed = 20 if RENDER in k else 1
    a = launches_of(cs, timed)
    out[k] = a
    print(f"### {k[:90]}\n")
    print("| counter | avg per launch |\n|---|---|")
    for c in sorted(a):
        print(f"| {c} | {a[c]:.6g} |")
    print()
json.dump(out, open(os.path.join(d, "pmc_avg.json"), "w"), indent=1)

for k, t in out.items():
    if RENDER not in k or "SQ_INSTS_VALU" not in t:
        continue
    cyc = t["GRBM_GUI_ACTIVE"] / 8.0                      # the counter sums the 8 XCDs
    rays = 2073600 * 204.8 * 1.37976                      # bench.py: rays per sample of this workload
    print("## KPIs of the render kernel (205-spp launch)\n")
    print(f"* VALU wave-instructions per ray: {t['SQ_INSTS_VALU'] / rays:.2f} (round 2: 29.7); SALU per VALU {t['SQ_INSTS_SALU'] / t['SQ_INSTS_VALU']:.2f}")
    print(f"* lanes active per VALU instruction: SQ_THREAD_CYCLES_VALU / (64 x SQ_INSTS_VALU) = {t['SQ_THREAD_CYCLES_VALU'] / (64 * t['SQ_INSTS_VALU']):.3f} (round 2: 0.59)")
    print(f"* VALUBusy by rocprof's formula 4 x SQ_ACTIVE_INST_VALU / (1024 SIMDs x cycles) = {4 * t['SQ_ACTIVE_INST_VALU'] / (1024 * cyc):.3f} (round 2: 0.96). The formula prices a wave64 VALU instruction at 4 cycles; gfx950's SIMDs are 32 lanes wide and issue one in 2 (MI355X_MICROARCH.md 'Wave scheduling'; tools/micro/valu_rate: 2.15 cycles with 4 waves per SIMD), so the share of VALU issue slots in use is {2 * t['SQ_INSTS_VALU'] / (1024 * cyc):.3f}")
    print(f"* wave time waiting on memory: SQ_WAIT_ANY / SQ_WAVE_CYCLES = {t['SQ_WAIT_ANY'] / t['SQ_WAVE_CYCLES']:.3f}; waiting for an issue slot: SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {t['SQ_WAIT_INST_ANY'] / t['SQ_WAVE_CYCLES']:.3f}")
    if "TCC_HIT_sum" in t and "TCC_MISS_sum" in t:
        print(f"* L2 hit rate {t['TCC_HIT_sum'] / (t['TCC_HIT_sum'] + t['TCC_MISS_sum']):.3f}; vector memory instructions {t.get('SQ_INSTS_VMEM_RD', 0) + t.get('SQ_INSTS_VMEM_WR', 0):.4g}, scalar {t.get('SQ_INSTS_SMEM', 0):.4g}")
    if avg_ms:
        print(f"* kernel trace: {avg_ms:.3f} ms average over all launches of the kernel")
    print()

# HBM-side traffic per launch: FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 reports half the bytes of 16-B-per-lane reads
# (MI355X_MICROARCH.md, "HBM"), so reads are doubled.
traffic = {}
for k, t in out.items():
    if RENDER in k and "FETCH_SIZE" in t:
        traffic[205] = t["FETCH_SIZE"] * 1024 * 2 + t.get("WRITE_SIZE", 0.0) * 1024
tr = collect("tr*")
for k, cs in tr.items():
    if RENDER in k and "FETCH_SIZE" in cs:
        a = launches_of(cs, 8)
        traffic[512] = a["FETCH_SIZE"] * 1024 * 2 + a.get("WRITE_SIZE", 0.0) * 1024
if traffic:
    res = {f"dragon_1920x1080_{s}spp": v for s, v in traffic.items()}
    if len(traffic) == 2:
        per = (traffic[512] - traffic[205]) / (512 - 204.8)
        res["dragon_1920x1080_model"] = {"fixed_bytes": traffic[205] - per * 204.8, "bytes_per_spp": per}
    res["_source"] = ("tools/profile_bench.sh: 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes) of k_megakernel_ab<false, true>, averaged over the timed launches of "
                      "`bench.py --steps 20 --warmup 5` (205 spp) and `--steps 8 --warmup 2` (512 spp), separate rocprofv3 --pmc passes; the model is the line "
                      "through the two. FETCH_SIZE counts the L2's fabric-side requests, Infinity-Cache hits included (MI355X_MICROARCH.md, HBM).")
    json.dump(res, open(os.path.join(d, "pmc_traffic.json"), "w"), indent=1)
    print("## fabric traffic per launch (2 x FETCH_SIZE + WRITE_SIZE)\n")
    for s, v in sorted(traffic.items()):
        print(f"* {s} spp: {v / 1e9:.3f} GB")
