#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=15 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log; tail -30 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step']); print('roofline',d['roofline']); print({k:v['avg_ms'] for k,v in d['kernels'].items()}); print('cfg2', d.get('other_configs'))
"
