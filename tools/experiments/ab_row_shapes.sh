for rep in 1 2; do for c in 2 5; do
  python tools/iter_rate.py --config $c --steps 200 --events --tag shipped 2>&1 | tail -1 | cut -c1-420
  PLSA_EXP_ROW_SHAPES=1 ENSTOP_AMD_LIB=$PWD/build/variants/libplsa_rowshapes.so python tools/iter_rate.py --config $c --steps 200 --events --tag row_half_lanes 2>&1 | tail -1 | cut -c1-420
done; done
