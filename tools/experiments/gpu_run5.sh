#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d["kernels"]
    print("   value %.2f it/s  ms/step %.3f n_gpus %d | "%(d["value"],d["ms_per_step"],d["n_gpus"]) + " ".join("%s=%.3f"%(k.replace("k_",""),v["avg_ms"]) for k,v in ks.items() if v["avg_ms"]>0.05))
    print("   e_step %.3f ms frac %.3f"%(d["roofline"]["avg_launch_ms"],d["roofline"]["frac"]))
except Exception as e:
    print("   parse failed",e); print(open(sys.argv[1].replace('.json','.err')).read()[-2500:])
PY
}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
echo "== cfg3 xcd split"; timeout 600 $B > gpurun_out/r5_xcd1.json 2> gpurun_out/r5_xcd1.err; summ gpurun_out/r5_xcd1.json
echo "== cfg3 no xcd split"; PLSA_XCD_SPLIT=0 timeout 600 $B > gpurun_out/r5_xcd0.json 2> gpurun_out/r5_xcd0.err; summ gpurun_out/r5_xcd0.json
echo "== cfg3 xcd split again"; timeout 600 $B > gpurun_out/r5_xcd1b.json 2> gpurun_out/r5_xcd1b.err; summ gpurun_out/r5_xcd1b.json
echo "== cfg2"; timeout 600 $B --config 2 > gpurun_out/r5_c2.json 2> gpurun_out/r5_c2.err; summ gpurun_out/r5_c2.json
echo "== force dist (torch + nccl, 1 rank)"; PLSA_BENCH_FORCE_DIST=1 timeout 900 $B > gpurun_out/r5_dist.json 2> gpurun_out/r5_dist.err; summ gpurun_out/r5_dist.json; tail -5 gpurun_out/r5_dist.err
echo "== torchrun 1 proc"; timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29544 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r5_torchrun.json 2> gpurun_out/r5_torchrun.err; summ gpurun_out/r5_torchrun.json
