#!/bin/bash
# config 3, final kernels: band size of the visiting order (documents per band; auto = 2048 = 512 KB of P(z|d) rows) and item length
mkdir -p gpurun_out/r04; out=gpurun_out/r04/band_seg_sweep_cfg3.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config 3 --steps 100 --reps 2 --tag "$*" 2>&1 | tail -1 | cut -c1-200 >> $out; }
run X=1
for b in 1024 4096 8192; do run PLSA_ORDER_BAND=$b; done
for s in 48 96 128; do run PLSA_COL_SEG=$s; done
run X=2
python - <<'PY'
import json
for ln in open("gpurun_out/r04/band_seg_sweep_cfg3.jsonl"):
    d = json.loads(ln); print("%-28s %7.1f it/s" % (d["tag"], d["iter_per_s"]))
PY
