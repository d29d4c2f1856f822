# round-3 evidence: official artefacts + counters + probes (one box, ~10 min)
bash tools/gpu_profile.sh 2>&1 | tail -25
bash tools/gpu_pmc.sh 2>&1 | tail -30
mkdir -p gpurun_out/r3
timeout 900 tools/sol/sol_probe 3 10 valu,gather,row,col,timeline,balance,order > gpurun_out/r3/sol_final.jsonl 2> gpurun_out/r3/sol_final.err; tail -2 gpurun_out/r3/sol_final.err
for seg in 8 16 32 0; do PLSA_E_ROWS=1 PLSA_E_SEG=$seg python tools/iter_rate.py --config 1 --estep --tag "cfg1 e-step rows seg=$seg" 2>/dev/null | tail -1; done | tee gpurun_out/r3/cfg1_estep.jsonl
python tools/iter_rate.py --config 1 --estep --tag "cfg1 e-step flat (default)" 2>/dev/null | tail -1 | tee -a gpurun_out/r3/cfg1_estep.jsonl
