#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
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
echo "== cfg3 UNR4 default"; timeout 600 $B > gpurun_out/r3_default.json 2> gpurun_out/r3_default.err; summ gpurun_out/r3_default.json
echo "== cfg3 UNR4 no item order"; PLSA_ITEM_ORDER=0 timeout 600 $B > gpurun_out/r3_noorder.json 2> gpurun_out/r3_noorder.err; summ gpurun_out/r3_noorder.json
for u in 2 8; do echo "== cfg3 UNR$u"; ENSTOP_AMD_LIB=$PWD/build/libplsa_unr$u.so timeout 600 $B > gpurun_out/r3_unr$u.json 2> gpurun_out/r3_unr$u.err; summ gpurun_out/r3_unr$u.json; done
for g in 4 16 32 64; do echo "== cfg3 grid mult $g"; PLSA_GRID_MULT=$g timeout 600 $B > gpurun_out/r3_g$g.json 2> gpurun_out/r3_g$g.err; summ gpurun_out/r3_g$g.json; done
for g in 16 64; do echo "== cfg3 UNR8 grid mult $g"; ENSTOP_AMD_LIB=$PWD/build/libplsa_unr8.so PLSA_GRID_MULT=$g timeout 600 $B > gpurun_out/r3_u8g$g.json 2> gpurun_out/r3_u8g$g.err; summ gpurun_out/r3_u8g$g.json; done
echo "== cfg3 materialised"; timeout 600 $B --schedule materialised > gpurun_out/r3_mat.json 2> gpurun_out/r3_mat.err; summ gpurun_out/r3_mat.json
for c in 2 1 5; do echo "== cfg$c"; timeout 900 python bench.py --config $c --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r3_c$c.json 2> gpurun_out/r3_c$c.err; summ gpurun_out/r3_c$c.json; done
