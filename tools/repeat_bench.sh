#!/bin/bash
# run-to-run spread of the default bench line: 8 fresh processes on one box
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --full --no-cpu-baseline --no-pmc --no-other-configs 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'run': $i, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'e_step_ms': d['roofline']['avg_launch_ms'], 'e_step_frac': d['roofline']['frac'], 'materialised': d['materialised_leg']['value'], 'p_fill_GBps': d['materialised_leg']['p_placement']['kept_fill_GBps']}))"
done
