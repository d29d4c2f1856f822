#!/bin/bash
# Round 4: document pass against real / all-hit / all-miss word maps, timings + rocprofv3 --pmc passes (counters in their own runs)
R=$PWD; O=$R/gpurun_out/r04/rowhitmiss; mkdir -p $O; export TMPDIR=/tmp
$R/tools/sol/sol_probe 3 5 rowhitmiss > $O/timings.jsonl 2> $O/timings.err
cd /tmp
run() { tag=$1; shift; timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d /tmp/rhm_$tag -o b -- $R/tools/sol/sol_probe 3 1 rowhitmiss > /dev/null 2> $O/pmc_$tag.err; }
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run tcp TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum
run ea TCC_EA0_RDREQ_sum
run fetch FETCH_SIZE
cd $R
python - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/rhm_*/**/*counter_collection.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_row_pass<plsa::Shape<8, 2, true, false>" in r["Kernel_Name"]]
    by = collections.defaultdict(list)
    for r in rows: by[r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for cn, v in by.items():
        v.sort()
        # launches in order: (warm-up + 1 timed) x 3 maps -> pairs
        vals = [x[1] for x in v]
        for i, name in enumerate(("real", "all_hit", "all_miss")):
            if 2 * i + 1 < len(vals): res[name][cn] = vals[2 * i + 1]
for f in glob.glob("/tmp/rhm_tcc/**/*kernel_trace.csv", recursive=True):
    rows = [r for r in csv.DictReader(open(f)) if "k_row_pass<plsa::Shape<8, 2, true, false>" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    for i, name in enumerate(("real", "all_hit", "all_miss")):
        if 2 * i + 1 < len(rows):
            r = rows[2 * i + 1]; res[name]["duration_ms_under_pmc"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
json.dump(res, open("gpurun_out/r04/rowhitmiss/pmc_summary.json", "w"), indent=1, sort_keys=True)
print(json.dumps(res, indent=0)[:1500])
PY
