#!/bin/bash
# PMC study of the small configs (serial schedule so that counters attribute to one kernel): where do the cycles go?
R=$PWD; mkdir -p $R/gpurun_out/r04; export TMPDIR=/tmp; cd /tmp
C=${C:-2}
run() { tag=$1; shift; PLSA_OVERLAP=${OV:-0} timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/spmc_$tag -o b -- python $R/tools/iter_rate.py --config $C --steps 20 --reps 1 > /dev/null 2> $R/gpurun_out/r04/spmc_$tag.err; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
run occ SQ_LEVEL_WAVES SQ_CYCLES SQ_IFETCH SQ_WAIT_INST_ANY
cd $R
python - <<'PY'
import csv, glob, collections, json, os
res = collections.defaultdict(dict)
for f in glob.glob("/tmp/spmc_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if "plsa::k_" not in k: continue
        short = k.split("(")[0].replace("void ", "").replace("plsa::", "")
        key = (short, row["Counter_Name"])
        agg[key][0] += 1; agg[key][1] += float(row["Counter_Value"])
    for (short, cn), (n, v) in agg.items():
        res[short][cn] = v / n
        res[short]["dispatches"] = n
out = "gpurun_out/r04/small_pmc_cfg%s_ov%s.json" % (os.environ.get("C", "2"), os.environ.get("OV", "0"))
json.dump(res, open(out, "w"), indent=1)
for short in sorted(res):
    if any(t in short for t in ("k_row_pass", "k_col_pass", "k_col_reduce", "k_row_reduce", "k_norm", "k_colsum")):
        print(short); print("   ", {k: round(v, 1) for k, v in sorted(res[short].items())})
PY
tail -2 gpurun_out/r04/spmc_*.err | head -30
