#!/bin/bash
# materialising E-step at the 20NG shape (config 1, 236 MB of P): traversal variants
mkdir -p gpurun_out/r04; out=gpurun_out/r04/estep_cfg1_variants.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config 1 --estep --reps 5 --tag "$*" 2>&1 | tail -1 | cut -c1-200 >> $out; }
run X=1
run PLSA_E_ROWS=1 PLSA_E_SEG=8
run PLSA_E_ROWS=1 PLSA_E_SEG=16
run PLSA_E_ROWS=1 PLSA_E_SEG=32
run PLSA_E_ROWS=1 PLSA_E_SEG=64
run PLSA_E_ROWS=1 PLSA_E_SEG=0
run PLSA_E_ROWS=0 PLSA_GRID_MULT=2
run PLSA_E_ROWS=0 PLSA_GRID_MULT=4
cat $out
