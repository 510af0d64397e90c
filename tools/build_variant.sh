#!/bin/bash
# Development aid: build libtracy_hip.so with extra -D flags into tools/variants/<name>.so (A/B runs pick it with TRACY_HIP_LIBRARY).
# usage: tools/build_variant.sh <name> [-DTRH_...=...] ...
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
H=tracy_amd/csrc/hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function -shared "$@" \
  -Iinclude $H/tracy_hip.hip $H/lbvh_build.hip $H/tile_order.hip $H/tracy_group.hip tracy_amd/csrc/host/bvh_builder.cpp -ldl -o tools/variants/$NAME.so
echo built tools/variants/$NAME.so
