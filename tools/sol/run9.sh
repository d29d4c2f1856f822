mkdir -p gpurun_out/r3
export TMPDIR=/tmp; R=$PWD; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr3 -o t -- python $R/tools/iter_rate.py --config 3 --steps 50 --reps 1 > /dev/null 2>&1
cd $R
python tools/trace_gaps.py /tmp/tr3
python - <<'PY'
import csv, glob
rows=[]
for f in glob.glob("/tmp/tr3/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n=r["Kernel_Name"]
        if "plsa::k_" not in n: continue
        short=n.replace("void ","").split("<")[0].replace("plsa::","")
        if "row_pass" in n and n.rstrip().endswith("true>(int const*, int const*, float const*, int, int const*, float const*, float const*, float const*, float*, float const*, float*, int, float, double*, int const*, int const*, int, long long, float*)"): short+="_LL"
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short))
rows.sort()
cols=[i for i,r in enumerate(rows) if r[2]=="k_col_pass"]
i0=cols[len(cols)-8]
t0=rows[i0][0]
for s,e,nm in rows[i0:i0+16]:
    print("%-20s start %9.1f us  end %9.1f us  dur %8.1f" % (nm,(s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3))
PY
