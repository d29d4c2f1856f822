#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python tools/iter_rate.py --config $CFG --estep --reps 2 --tag $tag 2>>gpurun_out/run12.err | cut -c1-110; }
CFG=3
run rows PLSA_E_ROWS=1 PLSA_ROW_ITEMS=0
run rows_items32 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=32
run rows_items64 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=64
run rows_items128 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=128
run rows_items256 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=256
CFG=2
run rows_items16 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=16
run rows_items24 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=24
run rows_items32 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=32
run rows_items48 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=48
tail -3 gpurun_out/run12.err
