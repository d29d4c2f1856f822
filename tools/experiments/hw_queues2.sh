#!/bin/bash
mkdir -p gpurun_out/r04; out=gpurun_out/r04/hw_queues_api_jobs.txt; : > $out
for q in 8 16 24; do
  echo "== API, GPU_MAX_HW_QUEUES=$q" >> $out
  ONLY=cfg4 JOBS=4,6,8 ENSTOP_AMD_CONCURRENT_MEMBERS_MAX=8 GPU_MAX_HW_QUEUES=$q python tools/ensemble_api_timing.py 2>&1 | grep cfg4 | cut -c60-200 >> $out
done
echo "== single fits, queues default vs 8 (iterations/s)" >> $out
for c in 3 1 2; do
  python tools/iter_rate.py --config $c --steps 200 --tag qdefault 2>&1 | tail -1 | cut -c1-110 >> $out
  GPU_MAX_HW_QUEUES=8 python tools/iter_rate.py --config $c --steps 200 --tag q8 2>&1 | tail -1 | cut -c1-110 >> $out
done
cat $out
