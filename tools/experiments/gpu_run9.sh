#!/bin/bash
export TMPDIR=/tmp
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --steps 6 --warmup 2 2>gpurun_out/r9.err | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); m=d['materialised_leg']
print('%-28s e_step %.3f ms frac %.3f | mat %.1f it/s | fused %.1f'%('$tag', d['roofline']['avg_launch_ms'], d['roofline']['frac'], m['value'], d['value']), m['p_placement']['kept_fill_GBps'])" || tail -5 gpurun_out/r9.err; }
ENSTOP_AMD_LIB=$PWD/build/libplsa_ps1.so timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "e_step or fit_vs_oracle or full_size" 2>&1 | tail -2
for rep in 1 2; do
run "nt (default)" X=1
run "sc1" ENSTOP_AMD_LIB=$PWD/build/libplsa_ps1.so
run "sc0 sc1" ENSTOP_AMD_LIB=$PWD/build/libplsa_ps2.so
run "sc1 nt" ENSTOP_AMD_LIB=$PWD/build/libplsa_ps3.so
done
