mkdir -p gpurun_out/r3
timeout 900 python bench.py --no-pmc > gpurun_out/r3/bench_c.json 2> gpurun_out/r3/bench_c.err; tail -3 gpurun_out/r3/bench_c.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3/bench_c.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms/step',d['ms_per_step']); print('roofline',d['roofline']['frac']); print('ensemble',json.dumps(d.get('ensemble'))[:600]); print('derived', d.get('ensemble_fits_per_min_from_iteration_rate')); print('cfg2', d.get('other_configs',{}).get('config2',{}).get('value'))
PY
python tools/ensemble_timing.py 2>&1 | tail -12
