#!/bin/bash
# the library's own default (constructor: GPU_MAX_HW_QUEUES=8 unless set) against the runtime default of 4, alternating
mkdir -p gpurun_out/r04; out=gpurun_out/r04/hw_queues_library_default.txt; : > $out
for rep in 1 2 3; do
  echo "== rep $rep: library default (8)" >> $out
  ONLY=cfg4 JOBS=4 python tools/ensemble_api_timing.py 2>&1 | grep cfg4 | cut -c60-200 >> $out
  echo "== rep $rep: GPU_MAX_HW_QUEUES=4 (runtime default)" >> $out
  ONLY=cfg4 JOBS=4 GPU_MAX_HW_QUEUES=4 python tools/ensemble_api_timing.py 2>&1 | grep cfg4 | cut -c60-200 >> $out
done
echo "== single fits (iterations/s), library default vs 4" >> $out
for rep in 1 2 3; do for c in 2 1 3; do
  python tools/iter_rate.py --config $c --steps 200 --tag q8 2>&1 | tail -1 | cut -c1-110 >> $out
  GPU_MAX_HW_QUEUES=4 python tools/iter_rate.py --config $c --steps 200 --tag q4 2>&1 | tail -1 | cut -c1-110 >> $out
done; done
cat $out
