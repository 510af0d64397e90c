#!/bin/bash
# Runs on the GPU box: the three pipelines side by side on one scene, counters summed over ALL kernels of a render call
# (the wavefront pipeline is ~5 kernels x ~50 iterations per call). usage: tools/pmc_pipelines.sh <scene> [spp]
set -u
SCENE=${1:-dragon}; SPP=${2:-64}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for PIPE in 0 3 1; do
  OUT=$ROOT/gpurun_out/pmcp_${SCENE}_$PIPE
  mkdir -p $OUT
  timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/prof_driver.py $SCENE $SPP 4 0 $PIPE > $OUT/stats.log 2>&1 || { echo "stats failed"; tail -3 $OUT/stats.log; exit 1; }
  i=0
  for PMC in "SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU" \
             "FETCH_SIZE TCC_HIT_sum" "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    i=$((i+1))
    timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/prof_driver.py $SCENE $SPP 4 0 $PIPE > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
  done
  python3 - "$OUT" "$PIPE" <<'PY'
import csv, glob, sys, collections
out, pipe = sys.argv[1], sys.argv[2]
launches = 6.0   # prof_driver: 4 + 2 warm-up render calls, all of the same size
tot = collections.defaultdict(float)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_megakernel" in r["Kernel_Name"] or "k_wf_" in r["Kernel_Name"]:
            tot[r["Counter_Name"]] += float(r["Counter_Value"])
ms = 0.0
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_megakernel" in r["Name"] or "k_wf_" in r["Name"]:
            ms += float(r["TotalDurationNs"]) / 1e6
            print(f"   {r['Name'][:60]:60s} calls {r['Calls']:>5s} total {float(r['TotalDurationNs'])/1e6:8.3f} ms")
t = {k: v / launches for k, v in tot.items()}
name = {"0": "megakernel (phases per wave, default)", "3": "megakernel_lanes (round 2)", "1": "wavefront (kernels per stage)"}[pipe]
gui = t.get("GRBM_GUI_ACTIVE", 0) / 8.0
print(f"pipeline {pipe} {name}: device time per render call {ms / launches:.3f} ms")
print(f"   VALU wave-instr {t.get('SQ_INSTS_VALU', 0):.4g}  SALU {t.get('SQ_INSTS_SALU', 0):.4g}  lanes per VALU instr {t.get('SQ_THREAD_CYCLES_VALU', 0) / max(1.0, 64 * t.get('SQ_INSTS_VALU', 1)):.3f}")
print(f"   VALUBusy (rocprof's 4-cycle formula: 4 * SQ_ACTIVE_INST_VALU / (1024 SIMDs * cycles)) {4 * t.get('SQ_ACTIVE_INST_VALU', 0) / max(1.0, 1024 * gui):.3f}; at the measured 2 cycles per wave64 VALU on gfx950's SIMD-32: {2 * t.get('SQ_INSTS_VALU', 0) / max(1.0, 1024 * gui):.3f}")
print(f"   wave time waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) {t.get('SQ_WAIT_ANY', 0) / max(1.0, t.get('SQ_WAVE_CYCLES', 1)):.3f}")
print(f"   L2: requests {t.get('TCC_REQ_sum', 0):.4g} hit rate {t.get('TCC_HIT_sum', 0) / max(1.0, t.get('TCC_HIT_sum', 0) + t.get('TCC_MISS_sum', 0)):.3f}; fabric traffic 2 x FETCH_SIZE + WRITE_SIZE = {(2 * t.get('FETCH_SIZE', 0) + t.get('WRITE_SIZE', 0)) * 1024 / 1e9:.3f} GB")
print(f"   vector memory instr {t.get('SQ_INSTS_VMEM_RD', 0) + t.get('SQ_INSTS_VMEM_WR', 0):.4g}, L1 accesses {t.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0):.4g}")
PY
done
