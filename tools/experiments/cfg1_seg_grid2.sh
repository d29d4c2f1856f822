mkdir -p gpurun_out/r04; out=gpurun_out/r04/cfg1_seg_grid2.txt; : > $out
run() { env "$@" python tools/iter_rate.py --config $C --steps 400 --reps 3 --tag "$*" 2>&1 | tail -1 | python -c "
import sys,re
ln=sys.stdin.read(); m=re.search(r'\"tag\": \"([^\"]+)\".*?\"iter_per_s\": ([\d.]+)', ln); print('$C', m.group(1), m.group(2))" >> $out; }
C=1
for rep in 1 2; do run X=1; for r in 28 32 40; do for s in 24 32 40; do run PLSA_ROW_SEG=$r PLSA_COL_SEG=$s; done; done; done
C=2
for rep in 1 2 3; do run X=1; run PLSA_COL_SEG=64; run PLSA_COL_SEG=48; done
cat $out
