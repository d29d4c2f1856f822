#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$PWD
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -x --tb=short 2>&1 | tail -30
O=gpurun_out/run6.jsonl; : > $O
for c in 1 2 3; do timeout 300 python tools/iter_rate.py --config $c --steps $([ $c = 3 ] && echo 50 || echo 200) --tag base >> $O 2>>gpurun_out/run6.err; done
cat $O | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['config'], d['tag'], d['ms_per_iter'], d['iter_per_s'])
"
cd /tmp
for c in 1 2; do
  rm -rf /tmp/tr$c; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$c -o t -- python $R/tools/iter_rate.py --config $c --steps 100 --reps 1 > /dev/null 2>>$R/gpurun_out/run6.err
  echo "config $c"; python $R/tools/trace_gaps.py /tmp/tr$c
done
cd $R; tail -3 gpurun_out/run6.err
