#!/bin/bash
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA 2>gpurun_out/r13.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s fused %.1f it/s %.3f ms/step | '%('$tag', d['value'], d['ms_per_step']) + ' '.join('%s=%.3f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items() if v['avg_ms']>0.02))" || tail -5 gpurun_out/r13.err; }
for c in 1 2; do EXTRA="--config $c"; for sg in 16 32 64 128 256; do run "cfg$c seg$sg" PLSA_COL_SEG=$sg; done; done
EXTRA="--config 1"; for sg in 16 32 64; do run "cfg1 serial seg$sg" PLSA_COL_SEG=$sg PLSA_OVERLAP=0; done
