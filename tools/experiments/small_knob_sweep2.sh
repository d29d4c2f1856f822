mkdir -p gpurun_out/r04; out=gpurun_out/r04/small_knob_sweep2.txt; : > $out
run() { env "$@" python tools/iter_rate.py --config $C --steps 400 --reps 3 --tag "$*" 2>&1 | tail -1 | python -c "
import sys,re
ln=sys.stdin.read(); m=re.search(r'\"tag\": \"([^\"]+)\".*?\"iter_per_s\": ([\d.]+)', ln); print('$C', m.group(1), m.group(2))" >> $out; }
C=1
for rep in 1 2; do run X=1; run PLSA_HEAVY_ITEMS=8; run PLSA_HEAVY_ITEMS=128; run PLSA_TAIL_PRIORITY=0; run PLSA_XCD_SPLIT=0; run PLSA_SORT_ROWS=0; run PLSA_ITEM_ORDER=0; done
C=2
for rep in 1 2; do run X=1; run PLSA_BALANCE=0; run PLSA_HEAVY_ITEMS=8; run PLSA_HEAVY_ITEMS=128; run PLSA_TAIL_PRIORITY=0; run PLSA_ITEM_ORDER=0; done
cat $out
