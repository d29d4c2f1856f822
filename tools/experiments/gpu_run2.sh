#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/pytest_gpu.log
tail -3 gpurun_out/pytest_gpu.log
python - <<'PY' 2>&1 | tee gpurun_out/stream_probe.txt
from enstop_amd.engine import Engine
e = Engine(0)
for kind, name in ((0, "fill nt"), (1, "fill plain"), (2, "copy")):
    for gb in (1, 8):
        print("%-10s %2d GB : %8.1f GB/s" % (name, gb, e.stream_bandwidth(gb << 30, kind, 5)))
PY
summ() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    ks=d["kernels"]
    print("   value %.2f it/s  ms/step %.3f | "%(d["value"],d["ms_per_step"]) + " ".join("%s=%.3f"%(k.replace("k_",""),v["avg_ms"]) for k,v in ks.items() if v["avg_ms"]>0.05))
    print("   e_step %.3f ms %.0f GB/s frac %.3f"%(d["e_step"]["avg_launch_ms"],d["e_step"]["achieved"],d["e_step"]["frac"]))
    if "cpu_baseline" in d: print("   cpu", d["cpu_baseline"])
except Exception as e:
    print("   parse failed",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
}
echo "== cfg3 default (with cpu baseline)"; timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; summ gpurun_out/bench_default.json
for v in 0 1 2 3; do echo "== cfg3 estep variant $v"; PLSA_ESTEP_VARIANT=$v timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_ev$v.json 2> gpurun_out/b_ev$v.err; summ gpurun_out/b_ev$v.json; done
for g in 4 16 32; do echo "== cfg3 grid mult $g (estep variant 0)"; PLSA_GRID_MULT=$g timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b_g$g.json 2> gpurun_out/b_g$g.err; summ gpurun_out/b_g$g.json; done
echo "== cfg3 unsorted rows"; PLSA_SORT_ROWS=0 timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b_unsorted.json 2> gpurun_out/b_unsorted.err; summ gpurun_out/b_unsorted.json
for sg in 64 256 512; do echo "== cfg3 col seg $sg"; PLSA_COL_SEG=$sg timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b_seg$sg.json 2> gpurun_out/b_seg$sg.err; summ gpurun_out/b_seg$sg.json; done
echo "== cfg2 / cfg1 default"; for c in 2 1; do timeout 600 python bench.py --config $c --no-cpu-baseline > gpurun_out/b_c$c.json 2> gpurun_out/b_c$c.err; summ gpurun_out/b_c$c.json; done
echo "== cfg3 materialised"; timeout 600 python bench.py --schedule materialised --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/b_mat.json 2> gpurun_out/b_mat.err; summ gpurun_out/b_mat.json
