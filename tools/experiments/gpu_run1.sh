#!/bin/bash
# first GPU session: parity tests, smoke, schedule comparison on config 2 and 3
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/pytest_gpu.log
echo "pytest exit: $?" >> gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
for sched in fused materialised fused+det materialised+det; do
  timeout 600 python bench.py --config 2 --steps 20 --warmup 3 --schedule $sched --no-cpu-baseline > gpurun_out/bench_c2_$sched.json 2> gpurun_out/bench_c2_$sched.err
done
for sched in fused materialised fused+det; do
  timeout 900 python bench.py --config 3 --steps 10 --warmup 2 --schedule $sched --no-cpu-baseline > gpurun_out/bench_c3_$sched.json 2> gpurun_out/bench_c3_$sched.err
done
tail -5 gpurun_out/pytest_gpu.log; cat gpurun_out/smoke.log | tail -3
for f in gpurun_out/bench_c*.json; do echo $f; python - "$f" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(" value",d["value"],"ms/step",d["ms_per_step"],"nnz",d["config"]["nnz"],"gen_s",d["generate_s"])
    for k,v in d["kernels"].items(): print("   ",k,v)
    print("  e_step",d["e_step"])
except Exception as e:
    print(" parse failed",e); print(open(sys.argv[1].replace('.json','.err')).read()[-1500:])
PY
done
