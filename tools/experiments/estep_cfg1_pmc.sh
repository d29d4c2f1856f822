#!/bin/bash
# HBM-side traffic of the materialising E-step at the 20NG shape (config 1): FETCH_SIZE and WRITE_SIZE in separate passes
R=$PWD; mkdir -p $R/gpurun_out/r04; export TMPDIR=/tmp; cd /tmp
for cn in FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum; do
  timeout 600 rocprofv3 --kernel-trace --pmc $cn --output-format csv -d /tmp/e1_$cn -o b -- python $R/tools/iter_rate.py --config 1 --estep --reps 2 > /dev/null 2> $R/gpurun_out/r04/e1_$cn.err
done
cd $R
python - <<'PY' | tee gpurun_out/r04/estep_cfg1_pmc.txt
import csv, glob, collections
res = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"):
    tot, n = 0.0, 0
    for f in glob.glob("/tmp/e1_%s/**/*counter_collection.csv" % cn, recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_e_step" in row["Kernel_Name"]:
                tot += float(row["Counter_Value"]); n += 1
    res[cn] = tot / max(n, 1)
    print(cn, "per launch:", res[cn], "(%d launches)" % n)
fetch, write = res["FETCH_SIZE"] * 1024 * 2, res["WRITE_SIZE"] * 1024      # KiB; FETCH_SIZE doubled on gfx950 (MI355X_MICROARCH.md)
print("traffic per launch: fetch %.1f MB + write %.1f MB = %.1f MB; algorithmic 263.1 MB" % (fetch / 1e6, write / 1e6, (fetch + write) / 1e6))
PY
