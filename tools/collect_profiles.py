#!/usr/bin/env python3
"""Turn the scratch output of tools/gpu_profile.sh (gpurun_out/) into the committed evidence under
profiles/: bench lines per config, the rocprofv3 --kernel-trace --stats table of the default bench
command (csv + a readable summary), the FETCH_SIZE / WRITE_SIZE PMC passes and the per-kernel HBM
traffic file bench.py reads (profiles/pmc_traffic.json).   usage: python tools/collect_profiles.py r01"""
import csv
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def short(name):
    name = name.replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    if "rocprim" in name or "hipcub" in name:
        return "rocprim radix sort / scan (corpus + structure build, untimed)"
    return name


def label(name):
    """bench.py's label of a hot kernel from its C++ name"""
    s = short(name).replace(" ", "")
    m = re.match(r"plsa::(k_\w+)<plsa::Shape<[\d,a-z]+>(?:,(\w+))?(?:,(\w+))?(?:,\w+)?>", s)
    if not m:
        return None
    k, a, b = m.groups()
    if k in ("k_e_step", "k_e_step_rows"):
        return "k_e_step"
    if k == "k_row_pass":
        return "k_row_pass<P>" if a == "true" else ("k_row_pass<fused,LL>" if b == "true" else "k_row_pass<fused>")
    if k == "k_col_pass":
        return "k_col_pass<P>" if a == "true" else "k_col_pass<fused>"
    return None


f = os.path.join(G, "bench_default_line.json")
if os.path.exists(f) and os.path.getsize(f) > 0:      # the line exactly as the default command printed it (round 6: compact)
    shutil.copy(f, os.path.join(P, "%s_bench_cfg3_1gpu_compact_line.json" % tag))
for src, dst in (("bench_cfg4.json", "%s_bench_cfg4_1gpu.json"), ("bench_cfg3_topical.json", "%s_bench_cfg3_topical_1gpu.json")):
    f = os.path.join(G, src)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        json.dump(last_json(f), open(os.path.join(P, dst % tag), "w"), indent=1)
f = os.path.join(G, "fuzz_parity_5x1000.txt")
if os.path.exists(f):
    shutil.copy(f, os.path.join(P, "%s_fuzz_parity_5x1000.txt" % tag))
f = os.path.join(G, "pytest_gpu.log")
if os.path.exists(f):
    shutil.copy(f, os.path.join(P, "%s_pytest_gpu.log" % tag))
for cfg, src in ((3, "bench_default.json"), (1, "bench_cfg1.json"), (2, "bench_cfg2.json"), (5, "bench_cfg5.json")):
    f = os.path.join(G, src)
    if os.path.exists(f):
        json.dump(last_json(f), open(os.path.join(P, "%s_bench_cfg%d_1gpu.json" % (tag, cfg)), "w"), indent=1)
for src, dst in (("bench_cfg2_world2_files.json", "%s_bench_cfg2_world2_on_one_gpu_files_test_mode.json"),):
    f = os.path.join(G, src)
    if os.path.exists(f) and os.path.getsize(f) > 0:
        json.dump(last_json(f), open(os.path.join(P, dst % tag), "w"), indent=1)
f = os.path.join(G, "bench_cfg2_world2_rccl_on_one_gpu.txt")
if os.path.exists(f):
    shutil.copy(f, os.path.join(P, "%s_bench_world2_rccl_refused_on_one_gpu.txt" % tag))
f = os.path.join(G, "bench_world2_one_rank_killed.err")
if os.path.exists(f):
    keep = [ln for ln in open(f) if ("enstop_amd rank" in ln or "exits at stage" in ln or ln.startswith("rc "))]
    with open(os.path.join(P, "%s_bench_world2_one_rank_killed_stage_lines.txt" % tag), "w") as o:
        o.write("PLSA_BENCH_FAIL_AT=1:timed python bench.py --gpus 2 --config 2 --steps 20 --exchange files   (one GPU box, host-file test mode)\n")
        o.writelines(keep)
f = os.path.join(G, "prof", "bench_under_rocprof.json")
if os.path.exists(f):
    json.dump(last_json(f), open(os.path.join(P, "%s_bench_cfg3_under_rocprofv3.json" % tag), "w"), indent=1)
f = os.path.join(G, "stream_probe.txt")
if os.path.exists(f):
    shutil.copy(f, os.path.join(P, "%s_stream_bandwidth_probe.txt" % tag))

f = os.path.join(G, "prof", "bench_kernel_stats.csv")
if os.path.exists(f):
    shutil.copy(f, os.path.join(P, "%s_rocprofv3_kernel_stats_bench_cfg3.csv" % tag))
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(P, "%s_rocprofv3_kernel_stats_bench_cfg3.summary.txt" % tag), "w") as o:
        o.write("rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline   (config 3, 1x MI355X)\n")
        o.write("HIP-event figures of the SAME process: profiles/%s_bench_cfg3_under_rocprofv3.json\n" % tag)
        o.write("%-62s %6s %12s %12s %12s %8s\n" % ("kernel", "calls", "avg_ms", "min_ms", "max_ms", "pct"))
        for r in rows[:40]:
            o.write("%-62s %6s %12.4f %12.4f %12.4f %8s\n" % (short(r["Name"])[:62], r["Calls"], float(r["AverageNs"]) / 1e6,
                                                               float(r["MinNs"]) / 1e6, float(r["MaxNs"]) / 1e6, r["Percentage"][:6]))

f = os.path.join(G, "prof", "pmc_summary.json")
if os.path.exists(f):
    pmc = json.load(open(f))
    dst = "%s_rocprofv3_pmc_fetch_write_cfg3.json" % tag
    json.dump(pmc, open(os.path.join(P, dst), "w"), indent=1)
    src_note = ("profiles/%s: (2 x FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes, FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 counts 128-B requests as 64 B)" % dst)
    per = {}
    for kind in ("fetch", "write"):
        for name, rec in pmc.get(kind, {}).get("per_kernel", {}).items():
            lab = label(name)
            if lab:
                per.setdefault(lab, {})[kind] = rec["avg_counter_value"]
    out = {}
    for lab, v in per.items():
        if "fetch" in v and "write" in v:
            out[lab] = {"fetch_size_kib_raw": v["fetch"], "write_size_kib_raw": v["write"],
                        "hbm_bytes_per_launch": int((2 * v["fetch"] + v["write"]) * 1024), "source": src_note}
    json.dump({"config3": out}, open(os.path.join(P, "pmc_traffic.json"), "w"), indent=1)
    for lab, v in out.items():
        print("%-24s %.2f GB per launch" % (lab, v["hbm_bytes_per_launch"] / 1e9))
