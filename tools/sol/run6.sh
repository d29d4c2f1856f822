mkdir -p gpurun_out/r3
timeout 2700 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest_gpu_b.log 2>&1; grep -E "passed|failed|error" gpurun_out/r3/pytest_gpu_b.log | tail -3; grep -n "Error\|FAILED\|assert " gpurun_out/r3/pytest_gpu_b.log | head -20
timeout 900 python bench.py --no-pmc > gpurun_out/r3/bench_b.json 2> gpurun_out/r3/bench_b.err; tail -3 gpurun_out/r3/bench_b.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_b.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step']); print('roofline',d['roofline']['frac']); print('ensemble',d.get('ensemble')); print('derived', d.get('ensemble_fits_per_min_from_iteration_rate')); print(d.get('other_configs',{}).get('config2',{}).get('value'))
PY
timeout 600 python bench.py --gpus 2 --config 2 --steps 20 --no-cpu-baseline --exchange files > gpurun_out/r3/bench_w2files.json 2> gpurun_out/r3/bench_w2files.err; tail -3 gpurun_out/r3/bench_w2files.err; python -c "
import json; d=json.loads(open('gpurun_out/r3/bench_w2files.json').read().strip().splitlines()[-1]); print(d['value'], d['ensemble'])"
