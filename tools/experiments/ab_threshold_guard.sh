#!/bin/bash
# same-box A/B of the threshold guard: PLSA_NOTHRESH=0 exact passes only, 1 guarded (default), 2 forced select-free
for rep in 1 2; do
for cfg in 2 1 3; do
  steps=400; [ $cfg = 3 ] && steps=60; [ $cfg = 1 ] && steps=1000
  for mode in 0 1 2; do
    PLSA_NOTHRESH=$mode python tools/iter_rate.py --config $cfg --steps $steps --reps 3 --tag "nothresh=$mode rep=$rep"
  done
done
done
