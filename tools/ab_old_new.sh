#!/bin/bash
# same-box A/B of two builds (build/variants/libplsa_old.so against the in-tree library), alternating, no per-kernel events
for rep in 1 2 3; do
  for c in ${CONFIGS:-3 1 2}; do
    ENSTOP_AMD_LIB=$PWD/build/variants/libplsa_old.so python tools/iter_rate.py --config $c --steps 200 --tag old 2>&1 | tail -1 | cut -c1-160
    python tools/iter_rate.py --config $c --steps 200 --tag new 2>&1 | tail -1 | cut -c1-160
  done
done
