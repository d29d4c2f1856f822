#!/bin/bash
R=$PWD; mkdir -p $R/gpurun_out/pmc; export TMPDIR=/tmp; cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "^\s*(Name|Counter_Name)\s*:\s*[A-Za-z0-9_]+" | awk '{print $NF}' | sort -u > $R/gpurun_out/pmc/counters.txt
wc -l $R/gpurun_out/pmc/counters.txt
run() { tag=$1; shift; timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/pmc_$tag -o b -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-ensemble > /dev/null 2> $R/gpurun_out/pmc/$tag.err; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
python - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in glob.glob("/tmp/pmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "plsa::k_" not in k: continue
        short = k.split("(")[0].replace("void ", "").replace("plsa::", "")
        key = (short, row["Counter_Name"])
        agg[key][0] += 1; agg[key][1] += float(row["Counter_Value"])
    for (short, cn), (n, v) in agg.items():
        res[short][cn] = v / n
json.dump(res, open("gpurun_out/pmc/summary.json", "w"), indent=1)
for short in sorted(res):
    if any(t in short for t in ("k_row_pass", "k_col_pass", "k_e_step", "k_loglik")):
        print(short); print("   ", {k: round(v, 1) for k, v in sorted(res[short].items())})
PY
tail -3 gpurun_out/pmc/*.err | head -40
