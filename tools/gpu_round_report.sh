#!/bin/bash
# Runs on the GPU box: the per-round result tables of BASELINE.md section 5. usage: tools/gpu_round_report.sh <tag>
TAG=${1:-r02}
cd "$(dirname "$0")/.."
CLI=tracy_amd/lib/tracy_hip_cli
if [ -z "$SKIP_CONFIGS" ]; then
  timeout -k 10 500 python tests/tools/run_configs.py gpurun_out/${TAG}_configs.json > gpurun_out/${TAG}_configs.log 2>&1 || { tail -5 gpurun_out/${TAG}_configs.log; exit 1; }
fi
{
  echo "== interactive host (tracy_hip_cli, dragon 1080p, 1024 spp in all)"
  for args in "-spu 1" "-spu 64" "-spu auto" "-spu auto -async"; do
    echo "-- $args"; timeout -k 10 120 $CLI -scene assets/data/scenes/dragon.scn -assets assets -kernel hip -width 1920 -height 1080 -spp 1024 $args 2>&1 | grep -i "msamples\|\*\*\*" || exit 1
  done
  echo "== launches of few samples (tools/gpu_small_spp.py)"
  timeout -k 10 120 python tools/gpu_small_spp.py dragon || exit 1
} > gpurun_out/${TAG}_interactive.txt 2>&1
cat gpurun_out/${TAG}_interactive.txt
