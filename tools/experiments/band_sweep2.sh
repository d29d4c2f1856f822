mkdir -p gpurun_out/r04; out=gpurun_out/r04/band_sweep2_cfg3.jsonl; : > $out
run() { env "$@" python tools/iter_rate.py --config $C --steps 100 --reps 2 --tag "$*" 2>&1 | tail -1 | cut -c1-200 >> $out; }
C=3
for rep in 1 2; do
run X=1
for b in 8192 12288 16384 24576 32768 65536; do run PLSA_ORDER_BAND=$b; done
done
C=5
run X=1
for b in 2048 4096 8192; do run PLSA_ORDER_BAND=$b; done
C=2
run X=1
for b in 8192 16384; do run PLSA_ORDER_BAND=$b; done
python - <<'PY'
import json
for ln in open("gpurun_out/r04/band_sweep2_cfg3.jsonl"):
    d = json.loads(ln); print(d["config"], "%-28s %8.1f it/s" % (d["tag"], d["iter_per_s"]))
PY
