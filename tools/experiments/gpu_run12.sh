#!/bin/bash
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA 2>gpurun_out/r12.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s fused %.1f it/s %.3f ms/step | '%('$tag', d['value'], d['ms_per_step']) + ' '.join('%s=%.3f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items() if v['avg_ms']>0.02))" || tail -5 gpurun_out/r12.err; }
for c in 1 2 3 5; do EXTRA="--config $c"; run "cfg$c" X=1; done
EXTRA="--config 1"; run "cfg1 serial" PLSA_OVERLAP=0
