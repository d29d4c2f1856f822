for rep in 1 2; do
  python tools/iter_rate.py --config 3 --steps 200 --events --tag lanes16x1 2>&1 | tail -1 | cut -c1-420
  PLSA_CHUNKS_PER_LANE=3 ENSTOP_AMD_LIB=$PWD/build/variants/libplsa_s82.so python tools/iter_rate.py --config 3 --steps 200 --events --tag lanes8x2 2>&1 | tail -1 | cut -c1-420
done
