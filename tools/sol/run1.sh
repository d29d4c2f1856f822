mkdir -p gpurun_out/r3
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -5
for seg in 256 128 96 64 48; do
  PLSA_COL_SEG=$seg python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag seg$seg
done
PLSA_BALANCE=0 PLSA_COL_SEG=256 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag nobal256
PLSA_BALANCE=0 PLSA_COL_SEG=64 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag nobal64
