#!/bin/bash
# banding only for frequent words
export TMPDIR=/tmp
echo "== tests (banding default)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== tests (tiny bands: 64 docs, hot_mult 1)"; PLSA_BAND_DOCS=64 PLSA_HOT_MULT=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "not full_size and not c_abi" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA 2>gpurun_out/r16.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s fused %.1f it/s %.3f ms | '%('$tag', d['value'], d['ms_per_step']) + ' '.join('%s=%.3f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items() if v['avg_ms']>0.02))" || tail -5 gpurun_out/r16.err; }
run "no banding" PLSA_BANDING=0
for hm in 8 32 128 512; do run "2MB hot_mult $hm" PLSA_HOT_MULT=$hm; done
for kb in 1024 4096 8192; do run "${kb}KB hot_mult 32" PLSA_BAND_KB=$kb; done
run "4MB hot_mult 128" PLSA_BAND_KB=4096 PLSA_HOT_MULT=128
