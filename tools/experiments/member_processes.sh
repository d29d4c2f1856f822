#!/bin/bash
mkdir -p gpurun_out/r04; out=gpurun_out/r04/member_processes.txt; : > $out
for P in 1 2 4 6; do
  start=$(python -c "import time; print(time.time() + 12)")
  for p in $(seq 1 $P); do python tools/experiments/member_processes.py 24 $start >> /tmp/mp_$P.txt 2>/dev/null & done
  wait
  python - <<PY >> $out
import json
rows=[json.loads(l) for l in open("/tmp/mp_$P.txt") if l.startswith("{")]
t0=min(r["t0"] for r in rows); t1=max(r["t1"] for r in rows); n=sum(r["members"] for r in rows)
print("processes $P: %d members in %.3f s = %.3f ms per member overall (%.0f fits/min); per process %s" % (n, t1-t0, (t1-t0)/n*1e3, n/(t1-t0)*60, [r["ms_per_member"] for r in rows]))
PY
done
cat $out
