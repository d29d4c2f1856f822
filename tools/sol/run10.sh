for i in 1 2; do
python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag prio 2>/dev/null | tail -1
PLSA_TAIL_PRIORITY=0 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag noprio 2>/dev/null | tail -1
ENSTOP_AMD_LIB=$PWD/enstop_amd/libplsa_alt.so python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag unrcol16 2>/dev/null | tail -1
done
for seg in 48 80 96; do PLSA_COL_SEG=$seg python tools/iter_rate.py --config 3 --steps 50 --reps 2 --tag seg$seg 2>/dev/null | tail -1; done
python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag c2prio 2>/dev/null | tail -1
PLSA_TAIL_PRIORITY=0 python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag c2noprio 2>/dev/null | tail -1
python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag c1prio 2>/dev/null | tail -1
PLSA_TAIL_PRIORITY=0 python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag c1noprio 2>/dev/null | tail -1
