mkdir -p gpurun_out/r04; out=gpurun_out/r04/small_knob_sweep.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config $C --steps 400 --reps 3 --tag "$*" 2>&1 | tail -1 | cut -c1-200 >> $out; }
C=2
for rep in 1 2; do run X=1; run PLSA_XCD_SPLIT=0; run PLSA_ORDER_BAND=1024; run PLSA_ORDER_BAND=2048; run PLSA_COL_SEG=32; run PLSA_COL_SEG=128; done
C=1
for rep in 1 2; do run X=1; run PLSA_ROW_SEG=32; run PLSA_ROW_SEG=128; run PLSA_COL_SEG=32; run PLSA_COL_SEG=16; run PLSA_ROW_ITEMS=0; done
python - <<'PY'
import json
for ln in open("gpurun_out/r04/small_knob_sweep.jsonl"):
    d = json.loads(ln); print(d["config"], "%-28s %8.1f it/s" % (d["tag"], d["iter_per_s"]))
PY
