#!/bin/bash
# hot-column tiles + reduce/normalise tail: parity, then rates
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -15
O=gpurun_out/run4.jsonl; : > $O
for c in 1 2; do
  PLSA_HOT=0 timeout 300 python tools/iter_rate.py --config $c --tag hot0 >> $O 2>>gpurun_out/run4.err
  PLSA_HOT=1 timeout 300 python tools/iter_rate.py --config $c --tag hot1 >> $O 2>>gpurun_out/run4.err
  PLSA_HOT=1 timeout 300 python tools/iter_rate.py --config $c --events --tag hot1_events >> $O 2>>gpurun_out/run4.err
done
PLSA_HOT=0 timeout 300 python tools/iter_rate.py --config 3 --steps 50 --events --tag hot0_events >> $O 2>>gpurun_out/run4.err
for hm in 4 8 16 32; do
  PLSA_HOT=1 PLSA_HOT_MIN=$hm timeout 300 python tools/iter_rate.py --config 3 --steps 50 --events --tag hot1_min${hm}_events >> $O 2>>gpurun_out/run4.err
done
PLSA_HOT=1 PLSA_HOT_MIN=8 PLSA_HOT_LDS_KB=32 timeout 300 python tools/iter_rate.py --config 3 --steps 50 --events --tag hot1_min8_lds32_events >> $O 2>>gpurun_out/run4.err
PLSA_HOT=0 timeout 300 python tools/iter_rate.py --config 3 --steps 50 --tag hot0 >> $O 2>>gpurun_out/run4.err
PLSA_HOT=1 timeout 300 python tools/iter_rate.py --config 3 --steps 50 --tag hot1 >> $O 2>>gpurun_out/run4.err
cat $O | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'], d['ll_last'], d.get('kernels',''))
"
tail -5 gpurun_out/run4.err
