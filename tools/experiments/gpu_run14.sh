#!/bin/bash
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA 2>gpurun_out/r14.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-18s fused %.1f it/s %.3f ms | e_step %.3f | '%('$tag', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms']) + ' '.join('%s=%.3f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items() if v['avg_ms']>0.1))" || tail -5 gpurun_out/r14.err; }
for rep in 1 2; do run "waves default" X=1; run "waves 8" ENSTOP_AMD_LIB=$PWD/build/libplsa_w8.so; done
EXTRA="--config 5"; run "cfg5 default" X=1; run "cfg5 waves 8" ENSTOP_AMD_LIB=$PWD/build/libplsa_w8.so
