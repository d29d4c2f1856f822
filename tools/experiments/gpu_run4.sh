#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d["kernels"]
    print("   value %.2f it/s  ms/step %.3f | "%(d["value"],d["ms_per_step"]) + " ".join("%s=%.3f"%(k.replace("k_",""),v["avg_ms"]) for k,v in ks.items() if v["avg_ms"]>0.05))
    print("   e_step %.3f ms %.0f GB/s frac %.3f"%(d["e_step"]["avg_launch_ms"],d["e_step"]["achieved"],d["e_step"]["frac"]))
except Exception as e:
    print("   parse failed",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
B="python bench.py --steps 10 --warmup 2 --no-cpu-baseline"
for g in 64 128 256 1024 100000; do echo "== cfg3 grid mult $g"; PLSA_GRID_MULT=$g timeout 600 $B > gpurun_out/r4_g$g.json 2> gpurun_out/r4_g$g.err; summ gpurun_out/r4_g$g.json; done
for g in 256 100000; do echo "== cfg3 UNR8 grid mult $g"; ENSTOP_AMD_LIB=$PWD/build/libplsa_unr8.so PLSA_GRID_MULT=$g timeout 600 $B > gpurun_out/r4_u8g$g.json 2> gpurun_out/r4_u8g$g.err; summ gpurun_out/r4_u8g$g.json; done
for g in 64 100000; do for c in 2 1; do echo "== cfg$c grid $g"; PLSA_GRID_MULT=$g timeout 900 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4_c${c}_g$g.json 2> gpurun_out/r4_c${c}_g$g.err; summ gpurun_out/r4_c${c}_g$g.json; done; done
