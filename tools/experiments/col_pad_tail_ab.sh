#!/bin/bash
# column pass: the remainder of an item as ONE padded batch of gathers (PLSA_COL_PAD_TAIL=1) against pairs (0); same box, alternating
mkdir -p gpurun_out/r05b; out=gpurun_out/r05b/col_pad_tail_ab.jsonl; : > $out
run() { cfg=$1; steps=$2; shift 2; env "$@" python tools/iter_rate.py --config $cfg --steps $steps --tag "$*" 2>&1 | tail -1 | cut -c1-300 >> $out; }
for rep in 1 2 3; do
  run 1 400 PLSA_COL_PAD_TAIL=0; run 1 400 PLSA_COL_PAD_TAIL=1
  run 2 300 PLSA_COL_PAD_TAIL=0; run 2 300 PLSA_COL_PAD_TAIL=1
done
run 3 50 PLSA_COL_PAD_TAIL=0; run 3 50 PLSA_COL_PAD_TAIL=1; run 3 50 PLSA_COL_PAD_TAIL=0
env PLSA_COL_PAD_TAIL=0 python tools/iter_rate.py --config 1 --steps 200 --events --tag "events pad0" 2>&1 | tail -1 | cut -c1-700 >> $out
env PLSA_COL_PAD_TAIL=1 python tools/iter_rate.py --config 1 --steps 200 --events --tag "events pad1" 2>&1 | tail -1 | cut -c1-700 >> $out
cat $out
