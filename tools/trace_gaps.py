#!/usr/bin/env python3
"""Idle time between the kernels of an EM iteration from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python tools/iter_rate.py --config 2 --steps 50 --reps 1
    python tools/trace_gaps.py DIR
Prints, for the steady-state iterations, the busy time per queue, the union busy time of the device, and
the wall time per iteration: wall - union = time in which NO kernel was running (launch / dependency gaps)."""
import csv
import glob
import os
import sys

rows = []
for f in glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"]
        if "plsa::k_" not in name:
            continue
        short = name.replace("void ", "").split("<")[0].replace("plsa::", "")
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short, r.get("Queue_Id", "?")))
rows.sort()
# steady state: everything between the first and last k_col_pass of the longest run
cols = [i for i, r in enumerate(rows) if r[2] == "k_col_pass"]
if len(cols) < 20:
    print("too few iterations in the trace"); sys.exit(1)
lo, hi = cols[len(cols) // 4], cols[-len(cols) // 4]
sel = rows[lo:hi]
n_iter = sum(1 for r in sel if r[2] == "k_col_pass")
wall = sel[-1][0] - sel[0][0]
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in sel:
    if cur_s is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
busy += cur_e - cur_s
per = {}
for s, e, nme, q in sel:
    per.setdefault(nme, [0, 0]); per[nme][0] += 1; per[nme][1] += e - s
print("iterations %d  wall/iter %.2f us  device-busy/iter %.2f us  idle/iter %.2f us (%.1f %%)"
      % (n_iter, wall / n_iter / 1e3, busy / n_iter / 1e3, (wall - busy) / n_iter / 1e3, 100.0 * (wall - busy) / wall))
for nme, (cnt, t) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-22s %6d launches  %8.2f us avg  %8.2f us per iteration" % (nme, cnt, t / cnt / 1e3, t / n_iter / 1e3))
