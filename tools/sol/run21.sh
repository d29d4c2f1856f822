for c in 2 1; do for i in 1 2; do
python tools/iter_rate.py --config $c --steps 200 --reps 3 --tag "config$c UNR=4" 2>/dev/null | tail -1 | cut -c1-120
ENSTOP_AMD_LIB=$PWD/enstop_amd/libplsa_alt.so python tools/iter_rate.py --config $c --steps 200 --reps 3 --tag "config$c UNR=8" 2>/dev/null | tail -1 | cut -c1-120
done; done
ENSTOP_AMD_LIB=$PWD/enstop_amd/libplsa_alt.so python tools/iter_rate.py --config 3 --steps 50 --reps 2 --tag "config3 UNR=8" 2>/dev/null | tail -1 | cut -c1-120
