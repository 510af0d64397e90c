"""Summarises a tools/profile_bench.sh output directory: kernel stats + per-kernel PMC averages."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

d = sys.argv[1]
print(f"# rocprofv3 summary ({os.path.basename(d)})\n")
for f in glob.glob(os.path.join(d, "stats", "**", "*kernel_stats.csv"), recursive=True):
    print("## kernel-trace --stats\n")
    print("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
    for r in csv.DictReader(open(f)):
        print(f"| {r['Name'][:70]} | {r['Calls']} | {float(r['TotalDurationNs'])/1e6:.3f} | {float(r['AverageNs'])/1e6:.3f} | {r['Percentage']} |")
    print()
agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("## PMC (average per dispatch; one rocprofv3 --pmc run per counter group)\n")
out = {}
for k, cs in agg.items():
    if "megakernel" not in k and "k_wf_extend" not in k and "k_wf_shade" not in k and "k_wf_generate" not in k:
        continue
    print(f"### {k[:90]}\n")
    print("| counter | avg per dispatch | dispatches |\n|---|---|---|")
    out[k] = {}
    for c, v in sorted(cs.items()):
        # the first dispatches are the instrumented pass and the warm-up; keep them out of the average when possible
        vv = v[2:] if len(v) > 3 else v
        a = sum(vv) / len(vv)
        out[k][c] = a
        print(f"| {c} | {a:.6g} | {len(vv)} |")
    print()
json.dump(out, open(os.path.join(d, "pmc_avg.json"), "w"), indent=1)

# HBM-side traffic per launch of the render kernel for bench.py's roofline.traffic: FETCH_SIZE / WRITE_SIZE are
# in KiB; gfx950 reports half the bytes of 16-B-per-lane reads (MI355X_MICROARCH.md, "HBM"), so reads are doubled.
for k, cs in out.items():
    if "megakernel<false" in k.replace(" ", "") and "FETCH_SIZE" in cs:
        traffic = cs["FETCH_SIZE"] * 1024 * 2 + cs.get("WRITE_SIZE", 0.0) * 1024
        print(f"\nHBM-side traffic per launch (2 x FETCH_SIZE + WRITE_SIZE): {traffic/1e9:.3f} GB")
        json.dump({"traffic_bytes_per_launch": traffic, "kernel": k[:80]}, open(os.path.join(d, "traffic.json"), "w"))
