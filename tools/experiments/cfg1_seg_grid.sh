mkdir -p gpurun_out/r04; out=gpurun_out/r04/cfg1_seg_grid.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config 1 --steps 400 --reps 3 --tag "$*" 2>&1 | tail -1 | cut -c1-200 >> $out; }
run X=1
for r in 16 24 32 48; do for s in 16 32 48 64; do run PLSA_ROW_SEG=$r PLSA_COL_SEG=$s; done; done
run X=2
python - <<'PY'
import json
for ln in open("gpurun_out/r04/cfg1_seg_grid.jsonl"):
    d = json.loads(ln); print("%-36s %8.1f it/s" % (d["tag"], d["iter_per_s"]))
PY
