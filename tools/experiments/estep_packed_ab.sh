#!/bin/bash
# round 5: packed-slot flat E-step (k_e_step_packed) against k_e_step at topic counts that leave lanes idle,
# and the bound of a word-segment-phased E-step at config 3 (same corpus shape with a vocabulary that fits one L2)
mkdir -p gpurun_out/r05b; out=gpurun_out/r05b/estep_packed_ab.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --estep --reps 5 --tag "$*" $ARGS 2>&1 | tail -1 | cut -c1-300 >> $out; }
ARGS="--config 1"; run PLSA_E_PACKED=0; run PLSA_E_PACKED=1; run PLSA_E_PACKED=0; run PLSA_E_PACKED=1
ARGS="--shape 18846,173762,2950000,10"; run PLSA_E_PACKED=0; run PLSA_E_PACKED=1
ARGS="--shape 18846,173762,2950000,27"; run PLSA_E_PACKED=0; run PLSA_E_PACKED=1
ARGS="--config 3"; run X=cfg3
ARGS="--shape 1000000,12500,100000000,64"; run X=vocab_fits_L2
ARGS="--shape 1000000,6250,100000000,64"; run X=vocab_half_L2
cat $out
