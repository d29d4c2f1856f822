for rep in 1 2 3; do
  python tools/iter_rate.py --config 3 --steps 200 --events --tag col16x1 2>&1 | tail -1 | cut -c1-420
  PLSA_EXP_COL_8X2=1 ENSTOP_AMD_LIB=$PWD/build/variants/libplsa_col82.so python tools/iter_rate.py --config 3 --steps 200 --events --tag col8x2_128threads 2>&1 | tail -1 | cut -c1-420
done
