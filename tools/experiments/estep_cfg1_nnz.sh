#!/bin/bash
# flat E-step at k = 20: time against the number of entries (intercept = fixed cost of the launch, slope = streaming rate)
mkdir -p gpurun_out/r05b; out=gpurun_out/r05b/estep_cfg1_nnz.jsonl; : > $out
run() { env PLSA_E_ROWS=0 python tools/iter_rate.py --estep --reps 5 --tag "$1" --shape $1 2>&1 | tail -1 | cut -c1-300 >> $out; }
run 4711,173762,737500,20
run 9423,173762,1475000,20
run 18846,173762,2950000,20
run 37692,173762,5900000,20
run 75384,173762,11800000,20
run 150768,173762,23600000,20
cat $out
