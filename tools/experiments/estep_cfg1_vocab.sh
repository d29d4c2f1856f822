#!/bin/bash
# config 1's E-step (82 us, 0.40 of the roofline on algorithmic bytes): how much of it are the P(w|z) rows that miss the L2s?
# Same documents and entries, vocabulary shrunk until the table fits one L2 (13.9 MB -> 3.2 MB -> 0.8 MB).
mkdir -p gpurun_out/r05b; out=gpurun_out/r05b/estep_cfg1_vocab.jsonl; : > $out
run() { python tools/iter_rate.py --estep --reps 5 --tag "$1" --shape $1 2>&1 | tail -1 | cut -c1-300 >> $out; }
run 18846,173762,2950000,20
run 18846,40000,2950000,20
run 18846,10000,2950000,20
run 18846,173762,2950000,32
run 18846,40000,2950000,32
cat $out
