#!/bin/bash
# usage: tools/prof_stats.sh <tag> [bench args]  -- kernel-trace --stats only
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_$TAG -- python3 $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > $ROOT/gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$ROOT/gpurun_out/prof_$TAG/**/*kernel_stats.csv",recursive=True)[0]
print("| kernel | calls | total ms | avg ms | % |\n|---|---|---|---|---|")
for r in csv.DictReader(open(f)):
    print("|", r["Name"][:60], "|", r["Calls"], "| %.3f | %.4f |"%(float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e6), r["Percentage"], "|")
PY
tail -1 $ROOT/gpurun_out/prof_$TAG.log | cut -c1-400
