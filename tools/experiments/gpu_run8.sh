#!/bin/bash
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['materialised_leg']
print('%-28s e_step %.3f ms frac %.3f | mat %.1f it/s | fused %.1f'%('$tag', d['roofline']['avg_launch_ms'], d['roofline']['frac'], m['value'], d['value']), m['p_placement']['kept_fill_GBps'])"; }
for rep in 1 2; do
run "default(UNR_E=4,g64)" X=1
run "UNR_E=1" ENSTOP_AMD_LIB=$PWD/build/libplsa_ue1.so
run "UNR_E=2" ENSTOP_AMD_LIB=$PWD/build/libplsa_ue2.so
run "UNR_E=8" ENSTOP_AMD_LIB=$PWD/build/libplsa_ue8.so
run "grid16" PLSA_GRID_MULT=16
run "grid32" PLSA_GRID_MULT=32
run "grid128" PLSA_GRID_MULT=128
run "grid100000" PLSA_GRID_MULT=100000
done
