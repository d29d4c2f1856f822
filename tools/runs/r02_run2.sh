#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q --durations=12 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/pytest_gpu.log; tail -60 gpurun_out/pytest_gpu.log
T0=$(date +%s)
timeout 1200 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "bench rc $? in $(( $(date +%s) - T0 )) s"; tail -c 600 gpurun_out/bench_default.err
python -c "
import json
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step']); print('roofline',d['roofline']); print('dom',d['roofline_dominant_fused']); print({k:v['avg_ms'] for k,v in d['kernels'].items()}); print('cfg2', d.get('other_configs')); print('cpu', d.get('cpu_baseline')); print('pmc', d.get('pmc_traffic'))
"
