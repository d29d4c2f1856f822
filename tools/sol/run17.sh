timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "not bench and not two_ranks" 2>&1 | grep -E "passed|failed" | tail -2
for c in 1 2; do for i in 1 2; do for p in 1 0; do
PLSA_PIPELINE=$p python tools/iter_rate.py --config $c --steps 200 --reps 3 --tag "config$c pipeline=$p" 2>/dev/null | tail -1 | cut -c1-120
done; done; done
python tools/ensemble_api_timing.py 2>/dev/null | head -4
PLSA_PIPELINE=0 python tools/ensemble_api_timing.py 2>/dev/null | head -4 | sed 's/^/pipeline=0 /'
