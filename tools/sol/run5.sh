timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | grep -E "passed|failed" | tail -2
for b in -1 0 1024 1536 4096; do
  PLSA_ORDER_BAND=$b python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag band$b 2>/dev/null | tail -1
done
for b in -1 0; do
PLSA_ORDER_BAND=$b python tools/iter_rate.py --config 5 --steps 10 --reps 2 --tag c5band$b 2>/dev/null | tail -1
PLSA_ORDER_BAND=$b python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag c2band$b 2>/dev/null | tail -1
PLSA_ORDER_BAND=$b python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag c1band$b 2>/dev/null | tail -1
done
PLSA_ORDER_BAND=2048 python tools/iter_rate.py --config 5 --steps 10 --reps 2 --tag c5band2048 2>/dev/null | tail -1
PLSA_ORDER_BAND=512 python tools/iter_rate.py --config 5 --steps 10 --reps 2 --tag c5band512 2>/dev/null | tail -1
