#!/bin/bash
# column-tail variants at the small configs
mkdir -p gpurun_out
export TMPDIR=/tmp
O=gpurun_out/run3_coop.jsonl; : > $O
for c in 1 2; do
  PLSA_COOP=0 timeout 300 python tools/iter_rate.py --config $c --tag coop0 >> $O 2>>gpurun_out/run3.err
  for b in 1 2 4 8; do PLSA_COOP=1 PLSA_COOP_BPC=$b timeout 300 python tools/iter_rate.py --config $c --tag coop1_bpc$b >> $O 2>>gpurun_out/run3.err; done
  PLSA_COOP=2 timeout 300 python tools/iter_rate.py --config $c --tag coop2 >> $O 2>>gpurun_out/run3.err
done
PLSA_COOP=0 timeout 300 python tools/iter_rate.py --config 1 --events --tag coop0_events >> $O 2>>gpurun_out/run3.err
PLSA_COOP=1 timeout 300 python tools/iter_rate.py --config 1 --events --tag coop1_events >> $O 2>>gpurun_out/run3.err
PLSA_COOP=1 timeout 300 python tools/iter_rate.py --config 2 --events --tag coop1_events >> $O 2>>gpurun_out/run3.err
cat $O | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'], d['ll_last'], d.get('kernels',''))
"
tail -5 gpurun_out/run3.err
timeout 1200 python -m pytest tests/test_hip_parity.py -m gpu -q -x 2>&1 | tail -5
