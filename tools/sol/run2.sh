mkdir -p gpurun_out/r3
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3/pytest_gpu_a.log 2>&1; tail -4 gpurun_out/r3/pytest_gpu_a.log
for seg in 256 128 64; do
  PLSA_COL_SEG=$seg python tools/iter_rate.py --config 5 --steps 10 --reps 2 --events --tag c5seg$seg 2>/dev/null | tail -1
done
python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag c2 2>/dev/null | tail -1
PLSA_BALANCE=1 python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag c2bal 2>/dev/null | tail -1
python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag c1 2>/dev/null | tail -1
