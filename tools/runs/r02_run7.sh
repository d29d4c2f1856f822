#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 2400 python -m pytest tests -m gpu -q --tb=short 2>&1 | tail -12
O=gpurun_out/run7.jsonl; : > $O
for c in 1 2 3; do timeout 300 python tools/iter_rate.py --config $c --steps $([ $c = 3 ] && echo 50 || echo 200) --tag base >> $O 2>>gpurun_out/run7.err; done
cat $O | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'])
"
cd /tmp
for c in 1 2; do
  rm -rf /tmp/tr$c; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$c -o t -- python $R/tools/iter_rate.py --config $c --steps 100 --reps 1 > /dev/null 2>>$R/gpurun_out/run7.err
  echo "config $c"; python $R/tools/trace_gaps.py /tmp/tr$c | tee $R/gpurun_out/trace_gaps_cfg$c.txt
done
cd $R
timeout 600 python tools/ensemble_timing.py 2>>gpurun_out/run7.err | tee gpurun_out/ensemble_timing.jsonl
