#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
PLSA_ESTEP_XCD=1 timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    m=d["materialised_leg"]; print("   value %.1f | e_step %.3f ms frac %.3f | mat %.2f it/s "%(d["value"],d["roofline"]["avg_launch_ms"],d["roofline"]["frac"],m["value"]) + " ".join("%s=%.3f"%(k.replace("k_",""),v["avg_ms"]) for k,v in m["kernels"].items() if v["avg_ms"]>0.2))
except Exception as e:
    print("   parse failed",e); print(open(sys.argv[1].replace('.json','.err')).read()[-2500:])
PY
}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
for rep in 1 2; do
for x in 0 1; do echo "== cfg3 ESTEP_XCD=$x"; PLSA_ESTEP_XCD=$x timeout 600 $B > gpurun_out/r7_x${x}_$rep.json 2> gpurun_out/r7_x${x}_$rep.err; summ gpurun_out/r7_x${x}_$rep.json; done
done
for c in 2 5; do for x in 0 1; do echo "== cfg$c ESTEP_XCD=$x"; PLSA_ESTEP_XCD=$x timeout 900 $B --config $c > gpurun_out/r7_c${c}x$x.json 2> gpurun_out/r7_c${c}x$x.err; summ gpurun_out/r7_c${c}x$x.json; done; done
