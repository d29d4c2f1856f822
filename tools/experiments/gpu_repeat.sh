#!/bin/bash
export TMPDIR=/tmp
for i in 1 2 3 4 5 6; do python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['materialised_leg']['kernels']
print('run $i value %.1f e_step %.3f ms frac %.3f rowP %.3f colP %.3f mat %.1f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], m['k_row_pass<P>']['avg_ms'], m['k_col_pass<P>']['avg_ms'], d['materialised_leg']['value']), d['materialised_leg']['p_placement'])"; done
for i in 1 2; do PLSA_PLACEMENT_CANDIDATES=1 python bench.py --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['materialised_leg']['kernels']
print('noshop $i value %.1f e_step %.3f ms frac %.3f rowP %.3f colP %.3f'%(d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], m['k_row_pass<P>']['avg_ms'], m['k_col_pass<P>']['avg_ms']))"; done
