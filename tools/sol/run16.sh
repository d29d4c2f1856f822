export TMPDIR=/tmp; R=$PWD
for c in 1 2; do
cd /tmp; rm -rf /tmp/tr$c
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$c -o t -- python $R/tools/iter_rate.py --config $c --steps 100 --reps 1 > /dev/null 2>&1
cd $R
echo "config $c"; python tools/trace_gaps.py /tmp/tr$c
python - <<PY
import csv, glob
rows=[]
for f in glob.glob("/tmp/tr$c/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "plsa::k_" not in n: continue
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n.replace("void ","").split("<")[0].split("(")[0].replace("plsa::",""), r.get("Queue_Id","?")))
rows.sort()
cols=[i for i,r in enumerate(rows) if r[2]=="k_col_pass"]
i0=cols[len(cols)-12]; t0=rows[i0][0]
for s,e,nm,q in rows[i0:i0+12]:
    print("  %-20s q%s start %7.1f us end %7.1f us dur %6.1f" % (nm,q,(s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3))
PY
done
