#!/bin/bash
# Development aid (GPU box): device builder variants against the host SAH builder: build time, SAH cost, box tests per ray,
# render time of a 64-spp launch. usage: tools/gpu_lbvh_ab.sh [scene[:subdivide] ...]
cd "$(dirname "$0")/.."
run() { echo -n "$scene subdivide $sub | $1 | "; shift; env TRH_SUBDIVIDE=$sub "$@" timeout -k 10 200 python tools/gpu_sweep.py child $scene $W $H 64 0 || exit 1; }
for sc in ${*:-dragon spheres helmet dragon:1 dragon:2}; do
  scene=${sc%%:*}; sub=0; [[ "$sc" == *:* ]] && sub=${sc##*:}
  W=1920; H=1080; [ "$scene" = helmet ] && { W=3840; H=2160; }
  run "host SAH" TRH_BUILDER=sah
  run "device (default): SAH top tree over 2-triangle clusters + SAH splits" TRH_BUILDER=lbvh
  run "device: SAH top tree over 8-triangle clusters (<= 2^15) + SAH splits" TRH_BUILDER=lbvh TRH_LBVH_CLUSTER=8 TRH_LBVH_MAXBITS=15
  run "device: SAH top tree over 8-triangle clusters (<= 2^15) + Morton splits (round 2's tree)" TRH_BUILDER=lbvh TRH_LBVH_CLUSTER=8 TRH_LBVH_MAXBITS=15 TRH_LBVH_SPLIT=morton
  run "device: SAH splits, no top tree" TRH_BUILDER=lbvh TRH_LBVH_TOP=0
  run "device: plain linear BVH" TRH_BUILDER=lbvh TRH_LBVH_TOP=0 TRH_LBVH_SPLIT=morton
done
