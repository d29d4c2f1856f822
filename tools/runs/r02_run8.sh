#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short -k "randomised or fit_vs" 2>&1 | tail -5
for c in 1 2; do timeout 300 python tools/iter_rate.py --config $c --steps 200 --tag base 2>>gpurun_out/run8.err | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'])
"; done
timeout 900 python tools/ensemble_jobs.py --config 1 --members 32 --jobs 1 2 3 4 6 8 2>>gpurun_out/run8.err | tee gpurun_out/ensemble_jobs_cfg1.jsonl
timeout 900 python tools/ensemble_jobs.py --config 2 --members 16 --jobs 1 2 4 2>>gpurun_out/run8.err | tee gpurun_out/ensemble_jobs_cfg2.jsonl
tail -3 gpurun_out/run8.err
