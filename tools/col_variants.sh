#!/bin/bash
# A/B of column-pass builds (build/variants/libplsa_*.so: -DPLSA_UNR_COL / -DPLSA_WAVES_COL) at one config:
#   gpurun -- 'bash tools/col_variants.sh 3 > gpurun_out/r04/col_variants_cfg3.jsonl'
cfg=${1:-3}
python tools/iter_rate.py --config $cfg --events --tag shipped 2>&1 | tail -1
for so in build/variants/libplsa_*.so; do
  ENSTOP_AMD_LIB=$PWD/$so python tools/iter_rate.py --config $cfg --events --tag $(basename $so .so) 2>&1 | tail -1
done
