python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag default 2>/dev/null | tail -1
PLSA_OVERLAP_FULL_LIMIT=1e13 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag fulloverlap 2>/dev/null | tail -1
PLSA_OVERLAP=0 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag nooverlap 2>/dev/null | tail -1
python tools/iter_rate.py --config 5 --steps 10 --reps 2 --tag c5 2>/dev/null | tail -1
python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag c2 2>/dev/null | tail -1
python tools/iter_rate.py --config 2 --steps 200 --reps 3 --events --tag c2ev 2>/dev/null | tail -1
