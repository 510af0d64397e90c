#!/bin/bash
# Runs on the GPU box: one rocprofv3 --pmc pass per argument (a quoted counter list) over a short bench.py run, prints the
# per-dispatch average of every counter for the megakernel. usage: tools/pmc_pass.sh <tag> "<counters>" ["<counters>" ...]
set -u
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for PMC in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $PMC --output-format csv -d $OUT/p$i -- python3 $ROOT/bench.py --spp-per-step 64 --steps 3 --warmup 1 --no-cpu-baseline --psnr-spp 0 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
  python3 - "$OUT/p$i" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_megakernel<false" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    # one row per (dispatch, counter); several rows of one dispatch (per XCD/SE) are summed by rocprofv3 already
    print(f"{k:36s} {sum(acc[k]) / len(acc[k]):.6g}  ({len(acc[k])} dispatches)")
PY
done
