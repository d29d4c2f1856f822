mkdir -p gpurun_out/r04
for c in 1 2; do for ov in 1 0; do
PLSA_OVERLAP=$ov python tools/iter_rate.py --config $c --steps 200 --events --tag "ov$ov" 
PLSA_OVERLAP=$ov python tools/iter_rate.py --config $c --steps 400 --tag "ov$ov-noev"
done; done > gpurun_out/r04/small_standalone.jsonl 2>&1
cat gpurun_out/r04/small_standalone.jsonl | cut -c1-1500
