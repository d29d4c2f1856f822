#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
for seed in 1 2 3; do timeout 1500 python tests/fuzz_parity.py 1000 $seed 2>>gpurun_out/run16.err | tail -12; done | tee gpurun_out/fuzz_parity_r02.txt
tail -3 gpurun_out/run16.err
