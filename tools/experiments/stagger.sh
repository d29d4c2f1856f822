#!/bin/bash
mkdir -p gpurun_out/r04; out=gpurun_out/r04/stagger_small_configs.jsonl; : > $out
for rep in 1 2 3; do for c in 1 2; do for st in 0 1; do
  PLSA_STAGGER=$st python tools/iter_rate.py --config $c --steps 400 --tag stagger$st 2>&1 | tail -1 | cut -c1-110 >> $out
done; done; done
PLSA_STAGGER=1 python tools/iter_rate.py --config 1 --steps 200 --events --tag stagger1_events 2>&1 | tail -1 >> $out
cat $out | cut -c1-600
