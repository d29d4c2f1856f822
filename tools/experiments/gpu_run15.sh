#!/bin/bash
# doc-band-major column items: parity + A/B vs the single-band layout, band size sweep
export TMPDIR=/tmp
echo "== tests (banding default)"; timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== tests (tiny bands: 64 docs)"; PLSA_BAND_DOCS=64 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "not full_size and not c_abi" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 20 --warmup 3 $EXTRA 2>gpurun_out/r15.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s fused %.1f it/s %.3f ms | '%('$tag', d['value'], d['ms_per_step']) + ' '.join('%s=%.3f'%(k.replace('k_',''),v['avg_ms']) for k,v in d['kernels'].items() if v['avg_ms']>0.02))" || tail -5 gpurun_out/r15.err; }
run "banded 2MB" X=1
run "no banding" PLSA_BANDING=0
for kb in 512 1024 3072 4096 8192; do run "band ${kb}KB" PLSA_BAND_KB=$kb; done
EXTRA="--config 5"; run "cfg5 banded" X=1; run "cfg5 no banding" PLSA_BANDING=0
EXTRA="--config 2"; run "cfg2 banded" X=1; run "cfg2 no banding" PLSA_BANDING=0
