#!/bin/bash
# Development aid (GPU box): tools/gpu_time.py under a list of env settings. usage: tools/gpu_env_sweep.sh "<scenes>" "A=1 B=2" "A=3" ...
cd "$(dirname "$0")/.."
SCENES=$1; shift
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg timeout -k 10 120 python tools/gpu_time.py $SCENES || exit 1
done
