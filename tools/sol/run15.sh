for i in 1 2 3; do
python tools/iter_rate.py --config 3 --steps 50 --reps 2 --tag new 2>/dev/null | tail -1 | cut -c1-110
ENSTOP_AMD_LIB=$PWD/enstop_amd/libplsa_alt.so python tools/iter_rate.py --config 3 --steps 50 --reps 2 --tag old 2>/dev/null | tail -1 | cut -c1-110
done
