#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_parity_at_scale.py -m gpu -q -x --tb=short -k "config4" 2>&1 | tail -5
for sg in 0 5; do for c in 1 2; do PLSA_SMALL_GRID=$sg timeout 300 python tools/iter_rate.py --config $c --steps 200 --tag small_grid$sg 2>>gpurun_out/run9.err | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'])
"; done; done
PLSA_SMALL_GRID=0 timeout 300 python tools/iter_rate.py --config 2 --steps 200 --tag small_grid0_again 2>>gpurun_out/run9.err | cut -c1-120
timeout 900 python tools/ensemble_api_timing.py 2>>gpurun_out/run9.err | tee gpurun_out/ensemble_api_timing.jsonl
tail -3 gpurun_out/run9.err
