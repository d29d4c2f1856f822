#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
PLSA_CHUNKS_PER_LANE=2 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu_cpl2.log; tail -2 gpurun_out/pytest_gpu_cpl2.log
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d["kernels"]
    print("   value %.2f it/s  ms/step %.3f | "%(d["value"],d["ms_per_step"]) + " ".join("%s=%.3f"%(k.replace("k_",""),v["avg_ms"]) for k,v in ks.items() if v["avg_ms"]>0.05))
    m=d["materialised_leg"]; print("   e_step %.3f ms frac %.3f | mat %.2f it/s "%(d["roofline"]["avg_launch_ms"],d["roofline"]["frac"],m["value"]) + " ".join("%s=%.3f"%(k.replace("k_",""),v["avg_ms"]) for k,v in m["kernels"].items() if v["avg_ms"]>0.2))
except Exception as e:
    print("   parse failed",e); print(open(sys.argv[1].replace('.json','.err')).read()[-2500:])
PY
}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
for rep in 1 2; do
echo "== cfg3 default (nt streams, 4 floats/lane)"; timeout 600 $B > gpurun_out/r6_a$rep.json 2> gpurun_out/r6_a$rep.err; summ gpurun_out/r6_a$rep.json
echo "== cfg3 no-nt build"; ENSTOP_AMD_LIB=$PWD/build/libplsa_nont.so timeout 600 $B > gpurun_out/r6_b$rep.json 2> gpurun_out/r6_b$rep.err; summ gpurun_out/r6_b$rep.json
echo "== cfg3 8 floats/lane"; PLSA_CHUNKS_PER_LANE=2 timeout 600 $B > gpurun_out/r6_c$rep.json 2> gpurun_out/r6_c$rep.err; summ gpurun_out/r6_c$rep.json
done
for c in 2 5; do
echo "== cfg$c 4 floats/lane"; timeout 900 $B --config $c > gpurun_out/r6_c${c}a.json 2> gpurun_out/r6_c${c}a.err; summ gpurun_out/r6_c${c}a.json
echo "== cfg$c 8 floats/lane"; PLSA_CHUNKS_PER_LANE=2 timeout 900 $B --config $c > gpurun_out/r6_c${c}b.json 2> gpurun_out/r6_c${c}b.err; summ gpurun_out/r6_c${c}b.json
done
