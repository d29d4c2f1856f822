#!/bin/bash
# official artefacts: gpu tests, default bench (with CPU baseline), rocprofv3 kernel stats of the same
# command, and HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs)
R=$PWD
mkdir -p $R/gpurun_out/prof
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/pytest_gpu.log; tail -2 gpurun_out/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/stats -o bench -- python $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof/bench_under_rocprof.json 2> $R/gpurun_out/prof/rocprof_stats.err
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/prof/pmc_fetch -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/rocprof_fetch.err
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/prof/pmc_write -o bench -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/prof/rocprof_write.err
cd $R
find /tmp/prof -type f | xargs ls -la | head -30
for f in $(find /tmp/prof/stats -name "*kernel_stats.csv"); do cp $f gpurun_out/prof/bench_kernel_stats.csv; done
python - <<'PY'
import csv, glob, collections, json
out = {}
for tag in ("fetch", "write"):
    for f in glob.glob("/tmp/prof/pmc_%s/**/*counter_collection.csv" % tag, recursive=True):
        agg = collections.defaultdict(lambda: [0, 0.0])
        rd = csv.DictReader(open(f))
        cols = rd.fieldnames
        for row in rd:
            k = row.get("Kernel_Name", "")
            agg[k][0] += 1; agg[k][1] += float(row.get("Counter_Value", 0))
        out[tag] = {"columns": cols, "per_kernel": {k: {"dispatches": n, "avg_counter_value": v / n} for k, (n, v) in agg.items()}}
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
            print("   %-5s %-70s n=%d avg=%.1f" % (tag, k[:70], n, v / n))
json.dump(out, open("gpurun_out/prof/pmc_summary.json", "w"), indent=1)
PY
head -20 gpurun_out/prof/bench_kernel_stats.csv
python -c "
import json; d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step']); print('roofline',d['roofline']); print('mat',d['materialised_leg']['value'], {k:v['avg_ms'] for k,v in d['materialised_leg']['kernels'].items()}); print('cpu',d.get('cpu_baseline'))"
