for seg in 8 16 32 64; do PLSA_COL_SEG=$seg python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag "c1 seg=$seg" 2>/dev/null | tail -1 | cut -c1-110; done
PLSA_BALANCE=1 python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag "c1 balance=1" 2>/dev/null | tail -1 | cut -c1-110
PLSA_ORDER_BAND=0 python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag "c1 band=0" 2>/dev/null | tail -1 | cut -c1-110
PLSA_XCD_SPLIT=0 python tools/iter_rate.py --config 1 --steps 200 --reps 3 --tag "c1 nosplit" 2>/dev/null | tail -1 | cut -c1-110
for seg in 16 32 64; do PLSA_COL_SEG=$seg python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag "c2 seg=$seg" 2>/dev/null | tail -1 | cut -c1-110; done
PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=32 python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag "c2 rowitems32" 2>/dev/null | tail -1 | cut -c1-110
PLSA_ROW_ITEMS=1 PLSA_ROW_SEG=64 python tools/iter_rate.py --config 2 --steps 200 --reps 3 --tag "c2 rowitems64" 2>/dev/null | tail -1 | cut -c1-110
