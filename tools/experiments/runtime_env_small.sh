#!/bin/bash
# runtime environment switches against the launch-chain-bound config 1 (and config 2): host wake-up mode, queue count
mkdir -p gpurun_out/r04; out=gpurun_out/r04/runtime_env_small.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config $C --steps 400 --tag "$*" 2>&1 | tail -1 | cut -c1-260 >> $out; }
for rep in 1 2; do for C in 1 2; do
run X=1
run HSA_ENABLE_INTERRUPT=0
run ROC_ACTIVE_WAIT_TIMEOUT=1000
run GPU_MAX_HW_QUEUES=2
run HIP_FORCE_DEV_KERNARG=1
done; done
python - <<'PY'
import json
for ln in open("gpurun_out/r04/runtime_env_small.jsonl"):
    try: d = json.loads(ln)
    except Exception: print(ln[:150]); continue
    print(d["config"], "%-36s %9.1f it/s" % (d["tag"], d["iter_per_s"]))
PY
