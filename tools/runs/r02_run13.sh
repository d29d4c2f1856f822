#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short -k "e_step or fit_vs or kernels_vs" 2>&1 | tail -3
run() { tag=$1; shift; env "$@" timeout 600 python tools/iter_rate.py --config $CFG --estep --reps 2 --tag $tag 2>>gpurun_out/run13.err | cut -c1-100; }
CFG=3
for sg in 0 48 64 80 96; do run seg$sg PLSA_E_SEG=$sg; done
CFG=2
for sg in 0 8 16 24 32 40; do run seg$sg PLSA_E_SEG=$sg; done
CFG=5
for sg in 0 32 64 96 128; do run seg$sg PLSA_E_SEG=$sg; done
CFG=1
run auto X=1
run rows16 PLSA_E_ROWS=1 PLSA_E_SEG=16
run rows24 PLSA_E_ROWS=1 PLSA_E_SEG=24
tail -3 gpurun_out/run13.err
