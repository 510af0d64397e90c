#!/bin/bash
# Development aid (GPU box): builder knobs (TRH_SAH_CT, TRH_BVH_ROTATE) against render time. usage: tools/gpu_bvh_ab.sh [scene ...]
cd "$(dirname "$0")/.."
for ct in ${AB_CT:-0.5 0.8}; do for rot in ${AB_ROT:-0 3}; do
  echo "== CT $ct rotate $rot"; TRH_SAH_CT=$ct TRH_BVH_ROTATE=$rot timeout -k 10 150 python tools/gpu_time.py ${*:-dragon} || exit 1
done; done
