#!/bin/bash
# config 2 document pass standalone (serial schedule, events on): does a finer work unit lift its 54 % VALU utilisation?
mkdir -p gpurun_out/r04
out=gpurun_out/r04/row_pass_granularity_cfg2.jsonl; : > $out
run() { tag=$1; shift; env "$@" PLSA_OVERLAP=0 python tools/iter_rate.py --config 2 --steps 100 --events --reps 2 --tag "$tag" 2>&1 | tail -1 >> $out; }
run base X=1
run nosort PLSA_SORT_ROWS=0
for seg in 16 32 64 128; do run items_seg$seg PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=$seg; done
for tag in base nosort; do :; done
# overlapped (shipped) schedule for the promising ones, no events
run2() { tag=$1; shift; env "$@" python tools/iter_rate.py --config 2 --steps 400 --tag "$tag" 2>&1 | tail -1 >> $out; }
run2 ov_base X=1
for seg in 16 32 64; do run2 ov_items_seg$seg PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=$seg; done
python - <<'PY'
import json
for ln in open("gpurun_out/r04/row_pass_granularity_cfg2.jsonl"):
    try: d = json.loads(ln)
    except Exception: print(ln[:200]); continue
    k = d.get("kernels", {})
    print("%-18s %8.1f it/s  row %s  reduce %s  col %s" % (d["tag"], d["iter_per_s"], k.get("k_row_pass<fused>"), k.get("k_row_reduce"), k.get("k_col_pass<fused>")))
PY
