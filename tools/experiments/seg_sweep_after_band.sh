mkdir -p gpurun_out/r04; out=gpurun_out/r04/seg_sweep_after_band.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config $C --steps 100 --reps 2 --tag "$*" 2>&1 | tail -1 | cut -c1-200 >> $out; }
C=3
for rep in 1 2; do run X=1; for s in 32 48 96 128; do run PLSA_COL_SEG=$s; done; done
C=5
run X=1; for s in 64 96; do run PLSA_COL_SEG=$s; done
python - <<'PY'
import json
for ln in open("gpurun_out/r04/seg_sweep_after_band.jsonl"):
    d = json.loads(ln); print(d["config"], "%-28s %8.1f it/s" % (d["tag"], d["iter_per_s"]))
PY
