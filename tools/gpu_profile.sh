#!/bin/bash
# official artefacts: gpu tests, default bench (with CPU baseline), rocprofv3 kernel stats of the same
# command, HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE in separate runs), stream probes
R=$PWD
mkdir -p $R/gpurun_out/prof
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -3
# the default command exactly as the driver runs it: the compact line on stdout, the full object in gpurun_out/bench_full_cfg3_n1.json
timeout 1500 python bench.py > gpurun_out/bench_default_line.json 2> gpurun_out/bench_default.err; tail -c 300 gpurun_out/bench_default.err
wc -c gpurun_out/bench_default_line.json; python -c "import json; print(json.dumps(json.load(open('gpurun_out/bench_full_cfg3_n1.json'))))" > gpurun_out/bench_default.json
for c in 2 1 5; do timeout 900 python bench.py --full --config $c --no-cpu-baseline --no-pmc > gpurun_out/bench_cfg$c.json 2> gpurun_out/bench_cfg$c.err; done
# the N = 2 launch path on this single-GPU box: bench.py spawns its own ranks; RCCL refuses two ranks on one
# device, so this is the host-file TEST MODE (labelled as such in the JSON) -- and the RCCL attempt must fail
timeout 600 python bench.py --gpus 2 --config 2 --steps 20 --no-cpu-baseline --exchange files > gpurun_out/bench_cfg2_world2_files.json 2> gpurun_out/bench_cfg2_world2_files.err
timeout 600 python bench.py --gpus 2 --config 2 --steps 20 --no-cpu-baseline > gpurun_out/bench_cfg2_world2_rccl_on_one_gpu.out 2> gpurun_out/bench_cfg2_world2_rccl_on_one_gpu.err; echo "rccl world-2 on one GPU: rc $? (expected non-zero), stdout bytes $(wc -c < gpurun_out/bench_cfg2_world2_rccl_on_one_gpu.out)" | tee gpurun_out/bench_cfg2_world2_rccl_on_one_gpu.txt
# multi-GPU first contact made cheap to read: rank 1 dies at the start of the timed region; rank 1's own line
# and rank 0's "terminated by the launcher" line name rank, device, stage, id file
PLSA_BENCH_FAIL_AT=1:timed timeout 600 python bench.py --gpus 2 --config 2 --steps 20 --no-cpu-baseline --no-pmc --exchange files > gpurun_out/bench_world2_one_rank_killed.out 2> gpurun_out/bench_world2_one_rank_killed.err; echo "rc $? (expected non-zero), stdout bytes $(wc -c < gpurun_out/bench_world2_one_rank_killed.out)" >> gpurun_out/bench_world2_one_rank_killed.err; grep -E "enstop_amd rank|exits at stage|^rc " gpurun_out/bench_world2_one_rank_killed.err
python - <<'PY' > gpurun_out/stream_probe.txt
from enstop_amd.engine import Engine
e = Engine(0)
for kind, name in ((0, "fill nt"), (1, "fill plain"), (2, "copy")):
    for gb in (1, 8):
        print("%-10s %2d GB : %8.1f GB/s" % (name, gb, e.stream_bandwidth(gb << 30, kind, 5)))
PY
cat gpurun_out/stream_probe.txt
cd /tmp
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
d=json.loads(open('gpurun_out/bench_default.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step']); print('roofline',d['roofline']); print('mat',d['materialised_leg']['value']); print('cpu',d.get('cpu_baseline'))
for c in (2,1,5):
    try:
        e=json.loads(open('gpurun_out/bench_cfg%d.json'%c).read().strip().splitlines()[-1]); print('cfg',c,e['value'],'it/s', e['roofline']['frac'], e['materialised_leg']['value'] if e['materialised_leg'] else None)
    except Exception as ex: print('cfg',c,'failed',ex)
PY
head -12 gpurun_out/prof/bench_kernel_stats.csv | cut -c1-160
# round 5: BASELINE configs[3] as its own command (32 members on the 20NG shape), the topical corpus with its counter traffic,
# the extended parity fuzz
timeout 600 python bench.py --full --config 4 --steps 200 --no-cpu-baseline > gpurun_out/bench_cfg4.json 2> gpurun_out/bench_cfg4.err
timeout 900 python bench.py --full --topics 64 --no-cpu-baseline --no-ensemble --no-other-configs > gpurun_out/bench_cfg3_topical.json 2> gpurun_out/bench_cfg3_topical.err
for s in 1 2 3 4 5; do timeout 900 python tests/fuzz_parity.py 1000 $s 2>&1 | tail -1; done | tee gpurun_out/fuzz_parity_5x1000.txt
