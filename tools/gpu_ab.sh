#!/bin/bash
# Development aid (GPU box): A/B of env knobs and tools/variants/*.so with tools/gpu_time.py. usage: tools/gpu_ab.sh [scene ...]
cd "$(dirname "$0")/.."
SCENES=${*:-dragon}
run() { echo "== $1"; shift; env "$@" timeout -k 10 120 python tools/gpu_time.py $SCENES || exit 1; }
run "default"
for t in ${AB_THRESH:-}; do run "refill_thresh $t" TRH_REFILL_THRESH=$t; done
for b in ${AB_BETA4:-}; do run "beta4 $b" TRH_SCHED_BETA4=$b; done
for v in tools/variants/*.so; do [ -e "$v" ] && run "variant $v" TRACY_HIP_LIBRARY=$PWD/$v; done
