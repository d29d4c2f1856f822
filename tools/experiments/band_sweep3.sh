mkdir -p gpurun_out/r04; out=gpurun_out/r04/band_sweep3.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config $C --steps 100 --reps 2 --tag "$*" 2>&1 | tail -1 | cut -c1-200 >> $out; }
C=3
for rep in 1 2; do for b in 2048 6144 8192 10240; do run PLSA_ORDER_BAND=$b; done; done
C=5
for rep in 1 2; do for b in 1024 3072 4096 6144; do run PLSA_ORDER_BAND=$b; done; done
python - <<'PY'
import json
for ln in open("gpurun_out/r04/band_sweep3.jsonl"):
    d = json.loads(ln); print(d["config"], "%-28s %8.1f it/s" % (d["tag"], d["iter_per_s"]))
PY
