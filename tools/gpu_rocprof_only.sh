R=$PWD; mkdir -p $R/gpurun_out/prof; export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -o bench -- python $R/bench.py --full --no-cpu-baseline --no-pmc --no-ensemble --no-other-configs > $R/gpurun_out/prof/bench_under_rocprof.json 2> $R/gpurun_out/prof/rocprof_stats.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-ensemble --no-other-configs > /dev/null 2> $R/gpurun_out/prof/rocprof_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-pmc --no-ensemble --no-other-configs > /dev/null 2> $R/gpurun_out/prof/rocprof_write.err
cd $R
for f in $(find /tmp/prof/stats -name "*kernel_stats.csv"); do cp $f gpurun_out/prof/bench_kernel_stats.csv; done
python - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("fetch", "write"):
    for f in glob.glob("/tmp/prof/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        for row in csv.DictReader(open(f)):
            k = row.get("Kernel_Name", "")
            if "plsa::" not in k: continue
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        out[tag] = {"per_kernel": {k: {"dispatches": n, "avg_counter_value": v / n} for k, (n, v) in agg.items()}}
json.dump(out, open("gpurun_out/prof/pmc_summary.json", "w"), indent=1)
PY
head -8 gpurun_out/prof/bench_kernel_stats.csv | cut -c1-120
