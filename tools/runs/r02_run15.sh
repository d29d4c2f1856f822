#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python tools/experiments/doc_reorder.py --config 3 2>gpurun_out/run15.err | tee gpurun_out/doc_reorder_cfg3.jsonl
tail -3 gpurun_out/run15.err
