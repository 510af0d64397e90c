#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 over EXACTLY the driver's bench command (bench.py --gpus 1 --steps 20 --warmup 5:
# 205-spp launches) -- kernel trace + separate --pmc passes -- and the fabric traffic at a second launch size, so that
# bench.py's roofline.traffic has a figure for any launch size (profiles/pmc_traffic.json: exact entries + linear model).
# usage: tools/profile_bench.sh <tag> [bench args, default: --gpus 1 --steps 20 --warmup 5]
set -u
TAG=${1:-r03}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS=${*:---gpus 1 --steps 20 --warmup 5}
BENCH="python3 $ROOT/bench.py --no-cpu-baseline --psnr-spp 0 $ARGS"
timeout -s KILL 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1 || { echo "stats pass failed"; tail -5 $OUT/stats.log; exit 1; }
i=0
for PMC in "FETCH_SIZE TCC_HIT_sum" \
           "WRITE_SIZE TCC_MISS_sum TCC_REQ_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_FMA_F32 SQ_INSTS_SMEM" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $PMC --output-format csv -d $OUT/pmc$i -- $BENCH > $OUT/pmc$i.log 2>&1 || { echo "pmc pass $i failed"; tail -5 $OUT/pmc$i.log; exit 1; }
  echo "pass $i done"
done
# the traffic of a second launch size (8 steps of 512 spp), for the model bytes = fixed + per_spp * spp
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 240 rocprofv3 --pmc $PMC --output-format csv -d $OUT/tr$i -- python3 $ROOT/bench.py --no-cpu-baseline --psnr-spp 0 --gpus 1 --steps 8 --warmup 2 > $OUT/tr$i.log 2>&1 || { echo "traffic pass $i failed"; tail -5 $OUT/tr$i.log; exit 1; }
done
python3 $ROOT/tools/summarize_prof.py $OUT > $OUT/summary.md && cat $OUT/summary.md
