set -e
for ct in 0.25 0.35 0.5 0.7 1.0; do echo -n "ct=$ct "; TRH_SAH_CT=$ct timeout -k 10 100 python tools/gpu_sweep.py child dragon 1920 1080 64 0; done
for t in 16 20 24 28 32; do echo -n "thresh=$t "; TRH_REFILL_THRESH=$t timeout -k 10 100 python tools/gpu_sweep.py child dragon 1920 1080 64 0; done
for b in 3 4 5 6 8 10; do echo -n "beta4=$b "; TRH_SCHED_BETA4=$b timeout -k 10 100 python tools/gpu_sweep.py child dragon 1920 1080 64 0; done
