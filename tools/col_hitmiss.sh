#!/bin/bash
# Round 4 column-pass experiment: shipped schedule against real / all-hit / all-miss row maps,
# timings first, then rocprofv3 --pmc passes (counters in their own runs, --kernel-trace only) over the same probe.
#   gpurun --timeout 1500 -- 'bash tools/col_hitmiss.sh'
R=$PWD; O=$R/gpurun_out/r04/hitmiss; mkdir -p $O; export TMPDIR=/tmp
$R/tools/sol/sol_probe 3 5 hitmiss > $O/timings.jsonl 2> $O/timings.err
cd /tmp
rocprofv3 -L 2>/dev/null > $O/list_avail.txt
grep -oE "(TCP|TA|TCC|TD|SQ)_[A-Za-z0-9_]+" $O/list_avail.txt | sort -u > $O/counters.txt
run() { tag=$1; shift; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/hm_$tag -o b -- $R/tools/sol/sol_probe 3 1 hitmiss > /dev/null 2> $O/pmc_$tag.err; }
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum
run tcp2 TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum
run ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
run tcp3 TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum
run sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAVES
run fetch FETCH_SIZE
run eamc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum
cd $R
python - <<'PY'
import csv, glob, collections, json, re
res = collections.defaultdict(dict)
for f in glob.glob("/tmp/hm_*/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        m = re.search(r"k_col_chunks<(\d+), (\d+), false, (\d+), (\d+)>", k)
        if not m or int(m.group(3)) < 10: continue
        key = ("mode%s_unr%s_tag%s_w%s" % m.groups(), row["Counter_Name"])
        agg[key][0] += 1; agg[key][1] += float(row["Counter_Value"])
    for (short, cn), (n, v) in agg.items():
        res[short][cn] = v / n
dur = collections.defaultdict(list)
for f in glob.glob("/tmp/hm_tcc/**/*kernel_trace.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        m = re.search(r"k_col_chunks<(\d+), (\d+), false, (\d+), (\d+)>", row["Kernel_Name"])
        if m and int(m.group(3)) >= 10:
            dur["mode%s_unr%s_tag%s_w%s" % m.groups()].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
for k, v in dur.items():
    res[k]["duration_ms_under_pmc"] = sum(v) / len(v)
json.dump(res, open("gpurun_out/r04/hitmiss/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(len(res), "kernel variants with counters")
PY
tail -2 $O/pmc_*.err | head -60
