mkdir -p gpurun_out/r3
OUT=gpurun_out/r3/hipgraph_on_off.jsonl; : > $OUT
for c in 1 2 3; do
  steps=200; [ $c = 3 ] && steps=50
  for g in 0 1; do
    PLSA_GRAPH=$g python tools/iter_rate.py --config $c --steps $steps --reps 3 --tag "config$c graph=$g" 2>/dev/null | tail -1 | tee -a $OUT
  done
done
for g in 0 1; do
  PLSA_GRAPH=$g python tools/ensemble_api_timing.py 2>/dev/null | tail -3 | sed "s/^/graph=$g /" | tee -a $OUT
done
PLSA_GRAPH=1 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "fit_vs_reference or fit_vs_oracle or randomised or earlystop" 2>&1 | grep -E "passed|failed" | tail -2
