#!/bin/bash
# hot-column tiles v2: parity, then a sweep at config 3 (and sanity at configs 1, 2, 5)
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x -k "hot or fit_vs_oracle or randomised or native or full_size" 2>&1 | tail -8
O=gpurun_out/run5.jsonl; : > $O
run() { tag=$1; shift; env "$@" timeout 300 python tools/iter_rate.py --config $CFG --steps $STEPS --reps 2 $EV --tag $tag >> $O 2>>gpurun_out/run5.err; }
CFG=3; STEPS=50; EV=--events
run hot0 PLSA_HOT=0
for lds in 16 32 48; do for hm in 8 16 32 64; do
  run hot_lds${lds}_min${hm} PLSA_HOT=2 PLSA_HOT_LDS_KB=$lds PLSA_HOT_MIN=$hm
done; done
EV=
run hot0_noev PLSA_HOT=0
run hot_lds32_min16_noev PLSA_HOT=2 PLSA_HOT_LDS_KB=32 PLSA_HOT_MIN=16
run hot_lds32_min32_noev PLSA_HOT=2 PLSA_HOT_LDS_KB=32 PLSA_HOT_MIN=32
CFG=2; STEPS=200
run hot0 PLSA_HOT=0
run hot2 PLSA_HOT=2
CFG=1
run hot0 PLSA_HOT=0
run hot2 PLSA_HOT=2
cat $O | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); k=d.get('kernels',{}); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'], d['ll_last'], {n:k[n] for n in k if 'col' in n or 'norm' in n})
"
tail -5 gpurun_out/run5.err
