timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "not fuzz and not bench and not two_ranks" 2>&1 | grep -E "passed|failed" | tail -2
timeout 600 python -m pytest tests/test_parity_at_scale.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for i in 1 2 3; do python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag dpp32 2>/dev/null | tail -1; done
python tools/iter_rate.py --config 5 --steps 10 --reps 2 --tag c5 2>/dev/null | tail -1
