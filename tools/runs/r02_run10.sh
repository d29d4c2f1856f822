#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short -k "hellinger" 2>&1 | tail -3
show() { python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'], d.get('kernels',''))
"; }
run() { tag=$1; shift; env "$@" timeout 300 python tools/iter_rate.py --config $CFG --steps 200 --reps 3 $EV --tag $tag 2>>gpurun_out/run10.err | show; }
CFG=2; EV=
run base X=1
run ritems PLSA_ROW_ITEMS=1
run ritems_seg32 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=32
run ritems_seg128 PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=128
run colseg16 PLSA_COL_SEG=16
run colseg64 PLSA_COL_SEG=64
run colseg128 PLSA_COL_SEG=128
run nooverlap PLSA_OVERLAP=0
EV=--events
run base_ev X=1
run ritems_ev PLSA_ROW_ITEMS=1
CFG=1; EV=
run base X=1
run noritems PLSA_ROW_ITEMS=0
run ritems_seg32 PLSA_ROW_SEG=32
run colseg32 PLSA_COL_SEG=32
run colseg8 PLSA_COL_SEG=8
tail -3 gpurun_out/run10.err
