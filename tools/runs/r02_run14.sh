#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
show() { python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); k=d.get('kernels',{}); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'], {n:k[n] for n in k if 'row' in n})
"; }
run() { tag=$1; shift; env "$@" timeout 300 python tools/iter_rate.py --config $CFG --steps 50 --reps 2 --events --tag $tag 2>>gpurun_out/run14.err | show; }
CFG=3
run base X=1
run ritems64 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=64
run ritems128 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=128
run ritems256 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=256
run ritems32 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=32
run nosort PLSA_SORT_ROWS=0
timeout 300 python tools/iter_rate.py --config 3 --estep --reps 2 --tag auto 2>>gpurun_out/run14.err | cut -c1-100
timeout 300 python tools/iter_rate.py --config 2 --estep --reps 2 --tag auto 2>>gpurun_out/run14.err | cut -c1-100
timeout 300 python tools/iter_rate.py --config 3 --steps 20 --reps 2 --flags materialised --events --tag materialised 2>>gpurun_out/run14.err | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'], d.get('kernels'))
"
tail -3 gpurun_out/run14.err
