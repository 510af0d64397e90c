#!/bin/bash
# Runs on the GPU box: kernel trace + separate rocprofv3 --pmc passes over tools/prof_driver.py; prints per-dispatch averages
# of the render kernel (the last `launches` dispatches). usage: [ENV=..] tools/pmc_quick.sh <tag> [driver args...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmcq_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_FMA_F32 SQ_INST_CYCLES_SALU" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TA_BUSY_avr TA_TA_BUSY_sum" \
           "FETCH_SIZE TCC_HIT_sum" \
           "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $ROOT/tools/prof_driver.py "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_megakernel" in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
for kn in acc:
    print("##", kn)
    for k in sorted(acc[kn]):
        v = [x[1] for x in sorted(acc[kn][k])][2:]   # skip the two warm-up launches
        if v:
            print(f"{k:36s} {sum(v) / len(v):.6g}  ({len(v)} dispatches)")
PY
