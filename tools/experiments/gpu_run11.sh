#!/bin/bash
export TMPDIR=/tmp
echo "== tests, default"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
echo "== tests, row items forced"; PLSA_ROW_ITEMS=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
echo "== tests, row items forced seg 4"; PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=4 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "not full_size and not large" 2>&1 | tail -4
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA 2>gpurun_out/r11.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s fused %.1f it/s %.3f ms/step | '%('$tag', d['value'], d['ms_per_step']) + ' '.join('%s=%.3f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items() if v['avg_ms']>0.02))" || tail -5 gpurun_out/r11.err; }
EXTRA="--config 1"
run "cfg1 auto" X=1
run "cfg1 no items" PLSA_ROW_ITEMS=0
for sg in 16 32 64 128; do run "cfg1 items seg$sg" PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=$sg; done
EXTRA="--config 2"; run "cfg2 auto" X=1; run "cfg2 items seg32" PLSA_ROW_ITEMS=1
EXTRA="--config 3"; run "cfg3 auto" X=1; run "cfg3 items seg32" PLSA_ROW_ITEMS=1
