#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" timeout 300 python tools/iter_rate.py --config $CFG --estep --reps 3 --tag $tag 2>>gpurun_out/run11.err | cut -c1-110; }
for CFG in 2 1; do
run flat PLSA_E_ROWS=0
run rows PLSA_E_ROWS=1 PLSA_ROW_ITEMS=0
run rows_items64 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1
run rows_items32 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=32
run rows_items128 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=128
run flat_grid32 PLSA_E_ROWS=0 PLSA_GRID_MULT=32
run flat_grid512 PLSA_E_ROWS=0 PLSA_GRID_MULT=512
run rows_items64_grid512 PLSA_E_ROWS=1 PLSA_ROW_ITEMS=1 PLSA_GRID_MULT=512
done
tail -3 gpurun_out/run11.err
