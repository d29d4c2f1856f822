#!/bin/bash
# config 1's fused iteration against the vocabulary size (same documents and entries): fewer rare words = fewer column items
mkdir -p gpurun_out/r05b; out=gpurun_out/r05b/cfg1_vocab_fused.jsonl; : > $out
for m in 173762 80000 40000 10000; do
  python tools/iter_rate.py --shape 18846,$m,2950000,20 --steps 400 --tag "m=$m" 2>&1 | tail -1 | cut -c1-400 >> $out
  python tools/iter_rate.py --shape 18846,$m,2950000,20 --steps 200 --events --tag "m=$m events" 2>&1 | tail -1 | cut -c1-900 >> $out
done
cat $out
