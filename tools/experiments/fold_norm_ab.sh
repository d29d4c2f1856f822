#!/bin/bash
# last stage of norm_pwz inside the column-reduce launch (PLSA_FOLD_NORM=1) against its own one-workgroup launch (0): same box, alternating
mkdir -p gpurun_out/r05b; out=gpurun_out/r05b/fold_norm_ab.jsonl; : > $out
run() { cfg=$1; steps=$2; shift 2; env "$@" python tools/iter_rate.py --config $cfg --steps $steps --tag "$*" 2>&1 | tail -1 | cut -c1-220 >> $out; }
for rep in 1 2 3 4; do
  run 1 400 PLSA_FOLD_NORM=0; run 1 400 PLSA_FOLD_NORM=1
  run 2 300 PLSA_FOLD_NORM=0; run 2 300 PLSA_FOLD_NORM=1
done
run 3 50 PLSA_FOLD_NORM=0; run 3 50 PLSA_FOLD_NORM=1
cat $out
