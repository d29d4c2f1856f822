bash tools/gpu_profile.sh 2>&1 | tail -14
mkdir -p gpurun_out/r3
for c in 1 2 3 5; do steps=200; [ $c = 3 ] && steps=50; [ $c = 5 ] && steps=10; python tools/iter_rate.py --config $c --steps $steps --reps 3 --tag "config $c, no per-kernel events" 2>/dev/null | tail -1; done | tee gpurun_out/r3/iter_rate_final.jsonl
python tools/ensemble_timing.py 2>/dev/null | grep -v host_mt | tee gpurun_out/r3/ensemble_timing_final.jsonl
python tools/ensemble_api_timing.py 2>/dev/null | tee gpurun_out/r3/ensemble_api_final.jsonl
