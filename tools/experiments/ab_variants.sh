#!/bin/bash
# same-box A/B of build/variants/libplsa_<name>.so against the in-tree library: VARIANTS="owner off32 both" CONFIGS="2 1 3"
mkdir -p gpurun_out/r04; out=gpurun_out/r04/ab_variants_${TAG:-x}.jsonl; : > $out
for rep in 1 2 3; do for c in ${CONFIGS:-2 1 3}; do
  python tools/iter_rate.py --config $c --steps 200 --tag base 2>&1 | tail -1 | cut -c1-120 >> $out
  for v in ${VARIANTS}; do
    ENSTOP_AMD_LIB=$PWD/build/variants/libplsa_$v.so python tools/iter_rate.py --config $c --steps 200 --tag $v 2>&1 | tail -1 | cut -c1-120 >> $out
  done
done; done
python - <<PY
import json, collections
r = collections.defaultdict(list)
for ln in open("$out"):
    try: d = json.loads(ln + ('' if ln.rstrip().endswith('}') else '"}'))
    except Exception:
        import re
        m = re.search(r'"tag": "(\w+)", "config": (\d+).*"iter_per_s": ([\d.]+)', ln)
        if m: r[(int(m.group(2)), m.group(1))].append(float(m.group(3)))
        continue
    r[(d["config"], d["tag"])].append(d["iter_per_s"])
for k in sorted(r): print(k, r[k])
PY
