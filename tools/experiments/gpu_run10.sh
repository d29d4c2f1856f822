#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA 2>gpurun_out/r10.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s fused %.1f it/s %.3f ms/step | '%('$tag', d['value'], d['ms_per_step']) + ' '.join('%s=%.2f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items() if v['avg_ms']>0.2))" || tail -5 gpurun_out/r10.err; }
for rep in 1 2; do
run "overlap" PLSA_OVERLAP=1
run "serial" PLSA_OVERLAP=0
done
EXTRA="--config 2"; run "cfg2 overlap" PLSA_OVERLAP=1; run "cfg2 serial" PLSA_OVERLAP=0
EXTRA="--config 5"; run "cfg5 overlap" PLSA_OVERLAP=1; run "cfg5 serial" PLSA_OVERLAP=0
EXTRA="--config 1"; run "cfg1 overlap" PLSA_OVERLAP=1; run "cfg1 serial" PLSA_OVERLAP=0
