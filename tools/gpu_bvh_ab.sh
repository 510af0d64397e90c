#!/bin/bash
# Development aid (GPU box): builder knobs against render time (TRH_SAH_CT for the host builder; TRH_LBVH_TOP /
# TRH_LBVH_CLUSTER / TRH_LBVH_DEPTH with AB_BUILDER=lbvh). usage: [AB_CT="0.5 0.8"] [AB_BUILDER=lbvh] tools/gpu_bvh_ab.sh [scene ...]
cd "$(dirname "$0")/.."
for ct in ${AB_CT:-0.5 0.8 1.0}; do
  echo "== C_traverse $ct builder ${AB_BUILDER:-sah}"; TRH_BUILDER=${AB_BUILDER:-sah} TRH_SAH_CT=$ct timeout -k 10 150 python tools/gpu_time.py ${*:-dragon} || exit 1
done
