#!/bin/bash
# config 1, serial schedule (one kernel at a time): the column pass against the number of resident workgroups per CU
mkdir -p gpurun_out/r05b; out=gpurun_out/r05b/cfg1_col_occupancy.jsonl; : > $out
for g in 1 2 3 4 5 6 8 12 0; do
  env PLSA_OVERLAP=0 PLSA_SMALL_GRID=$g python tools/iter_rate.py --config 1 --steps 200 --events --tag "serial small_grid=$g" 2>&1 | tail -1 | cut -c1-700 >> $out
done
cat $out
