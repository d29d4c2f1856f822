for i in 1 2; do
python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag depth16 2>/dev/null | tail -1
ENSTOP_AMD_LIB=$PWD/enstop_amd/libplsa_alt.so python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag depth4 2>/dev/null | tail -1
done
python tools/iter_rate.py --config 3 --steps 50 --reps 3 --tag depth16_noevents 2>/dev/null | tail -1
ENSTOP_AMD_LIB=$PWD/enstop_amd/libplsa_alt.so python tools/iter_rate.py --config 3 --steps 50 --reps 3 --tag depth4_noevents 2>/dev/null | tail -1
PLSA_HEAVY_ITEMS=128 python tools/iter_rate.py --config 3 --steps 50 --reps 3 --tag heavy128 2>/dev/null | tail -1
PLSA_HEAVY_ITEMS=8 python tools/iter_rate.py --config 3 --steps 50 --reps 3 --tag heavy8 2>/dev/null | tail -1
timeout 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "two_ranks or rccl or big or fuzz" 2>&1 | grep -E "passed|failed" | tail -3
