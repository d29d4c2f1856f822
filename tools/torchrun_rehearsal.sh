#!/bin/bash
# Dress rehearsal of the driver's launch shapes on a single-GPU box (round 4: token-carrying rendezvous file, stage lines).
#   gpurun -- 'bash tools/torchrun_rehearsal.sh > gpurun_out/r04/torchrun_rehearsal.txt 2>&1'
B="--steps 5 --warmup 1 --config 2 --no-cpu-baseline --no-pmc"
L="python -m torch.distributed.run --nnodes=1 --master-addr 127.0.0.1"
echo "== N=1 under torchrun"
$L --nproc-per-node 1 --master-port 29511 bench.py --gpus 1 $B 2>/tmp/e1 | cut -c1-200; echo "rc ${PIPESTATUS[0]}"
echo "== N=2 under torchrun, RCCL on ONE GPU (must fail: duplicate GPU; every rank names its stage)"
$L --nproc-per-node 2 --master-port 29512 bench.py --gpus 2 $B > /tmp/o2 2>/tmp/e2; echo "rc $? stdout bytes $(wc -c < /tmp/o2)"
grep -E "enstop_amd rank|Duplicate GPU" /tmp/e2 | cut -c1-400 | head -8
echo "== N=2 under torchrun, --exchange files (labelled test mode)"
$L --nproc-per-node 2 --master-port 29513 bench.py --gpus 2 $B --exchange files 2>/tmp/e3 | cut -c1-260; echo "rc ${PIPESTATUS[0]}"
grep -E "enstop_amd rank" /tmp/e3 | head -3
echo "== N=2 under torchrun, files, rank 1 dies in the ensemble leg (survivor must report its stage when torchrun tears the job down)"
PLSA_BENCH_FAIL_AT=1:ensemble $L --nproc-per-node 2 --master-port 29514 bench.py --gpus 2 $B --exchange files > /tmp/o4 2>/tmp/e4; echo "rc $? stdout bytes $(wc -c < /tmp/o4)"
grep -E "enstop_amd rank|exits at stage" /tmp/e4 | cut -c1-400 | head -6
