#!/bin/bash
# do the ensemble members of several contexts on one GPU serialise on the 4 default hardware queues?
mkdir -p gpurun_out/r04; out=gpurun_out/r04/hw_queues_member_scaling.txt; : > $out
for q in default 8 16; do
  echo "== GPU_MAX_HW_QUEUES=$q" >> $out
  if [ $q = default ]; then python tools/experiments/member_threads_scaling.py 2>&1 | grep -E '"bif"|"i"' >> $out
  else GPU_MAX_HW_QUEUES=$q python tools/experiments/member_threads_scaling.py 2>&1 | grep -E '"bif"|"i"' >> $out; fi
done
for q in default 8 16; do
  echo "== API, GPU_MAX_HW_QUEUES=$q" >> $out
  if [ $q = default ]; then python tools/ensemble_api_timing.py 2>&1 | grep cfg4 >> $out
  else GPU_MAX_HW_QUEUES=$q python tools/ensemble_api_timing.py 2>&1 | grep cfg4 >> $out; fi
done
cat $out | cut -c1-220
