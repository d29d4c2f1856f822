for span in 0 1024 2048 4096 8192; do
  PLSA_COL_SPAN=$span python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag span$span 2>/dev/null | tail -1
done
PLSA_COL_SPAN=2048 PLSA_COL_SEG=32 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag span2048min32 2>/dev/null | tail -1
PLSA_COL_SPAN=4096 PLSA_COL_SEG=48 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag span4096min48 2>/dev/null | tail -1
PLSA_COL_SPAN=2048 PLSA_COL_SEG_MAX=512 python tools/iter_rate.py --config 3 --steps 50 --reps 2 --events --tag span2048max512 2>/dev/null | tail -1
for span in 0 2048 8192; do
PLSA_COL_SPAN=$span python tools/iter_rate.py --config 5 --steps 10 --reps 2 --tag c5span$span 2>/dev/null | tail -1
done
