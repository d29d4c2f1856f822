#!/usr/bin/env python3
"""Speed-of-light table of the hot kernels (round 3): per kernel, the rocprofv3 average duration, the counters of
profiles/r03_rocprofv3_pmc_sq_tcc_cfg3.json and the ceilings measured by tools/sol/sol_probe on the same kind of box
(profiles/r03_sol_probe.jsonl) -> profiles/r03_speed_of_light.{json,md}.       usage: python tools/sol/speed_of_light.py
Ceilings (measured, not nominal):
  VALU issue        wave-instructions per SIMD per microsecond of a saturating v_fma_f32 loop (8 waves per SIMD)
  L2-hit gathers    random 256-byte rows out of a 2 MB table (fits every XCD's L2): rows/ns, 2 lines of 128 B each
  L2-miss gathers   random 256-byte rows out of a 25 MB table (the P(w|z) table: Infinity Cache) / a 256 MB table (the
                    P(z|d) table) / a 2 GB table (HBM)
The gather pipeline of a CU serves hits and misses one after the other: the additive model
  t_model = hit_rows / rate_hit + miss_rows / rate_miss          (rows = L2 requests / 2)
is the bound the two fused passes are compared with, next to the same traversal with all arithmetic removed."""
import csv
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
P = os.path.join(ROOT, "profiles")
pmc = json.load(open(os.path.join(P, "r03_rocprofv3_pmc_sq_tcc_cfg3.json")))
probe = [json.loads(l) for l in open(os.path.join(P, "r03_sol_probe.jsonl")) if l.strip()]
stats = {}
for r in csv.DictReader(open(os.path.join(P, "r03_rocprofv3_kernel_stats_bench_cfg3.csv"))):
    stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]) / 1e6)

valu = next(p for p in probe if p.get("test") == "valu" and p["op"] == "v_fma_f32")["wave_instr_per_simd_per_us"]
rate = {}
for p in probe:
    if p.get("test") == "random_row_gather" and p["rows_in_flight"] == 8:
        rate[p["table_mb"]] = p["rows_per_ns"]
floor = {}
for p in probe:
    if p.get("test") == "pass" and "gather-only, 8" in p["kernel"]:
        floor["row" if p["kernel"].startswith("row") else "col_unbalanced_256"] = p["ms"]
for p in probe:
    if p.get("test") == "col_order" and p["band"] == 2048 and p["round"] == 3:
        floor["col"] = p["ms_gather_only"]
        floor["col_full_in_probe"] = p["ms"]


def find(prefix, *flags):
    for name, (calls, ms) in stats.items():
        if prefix in name and all(f in name for f in flags):
            return name, calls, ms
    return None, 0, None


rows = []
SIMDS = 1024
for label, key, statkey, table_mb, floor_key in (
        ("k_row_pass<fused> (document pass)", "k_row_pass<Shape<16, 1, true>, false, false>", ("k_row_pass<", "true>, false, false>"), 25, "row"),
        ("k_col_pass<fused> (column pass)", "k_col_pass<Shape<16, 1, true>, false, false>", ("k_col_pass<", "true>, false, false>"), 256, "col"),
        ("k_e_step (materialising E-step)", "k_e_step_rows<Shape<16, 1, true> >", ("k_e_step_rows<",), 25, None)):
    c = pmc[key]
    _, calls, ms = find(*statkey)
    us = ms * 1e3
    hit_rows, miss_rows = c["TCC_HIT_sum"] / 2.0, c["TCC_MISS_sum"] / 2.0
    t_model = (hit_rows / rate[2] + miss_rows / rate[table_mb]) / 1e6            # ms
    entry = {
        "kernel": label, "rocprofv3_avg_ms": round(ms, 4), "launches_in_profile": calls,
        "effective_clock_GHz": round(c["GRBM_GUI_ACTIVE"] / 8.0 / us / 1e3, 3),
        "valu_wave_instr_per_launch": int(c["SQ_INSTS_VALU"]),
        "valu_issue_per_simd_per_us": round(c["SQ_INSTS_VALU"] / SIMDS / us, 1),
        "valu_issue_ceiling_per_simd_per_us": valu,
        "valu_issue_frac": round(c["SQ_INSTS_VALU"] / SIMDS / us / valu, 3),
        "l2_requests_per_launch": int(c["TCC_REQ_sum"]), "l2_hit_rate": round(c["TCC_HIT_sum"] / c["TCC_REQ_sum"], 3),
        "l2_requests_G_per_s": round(c["TCC_REQ_sum"] / us / 1e3, 1),
        "l2_request_ceiling_G_per_s": round(2 * rate[2], 1),
        "l2_miss_bytes_GB": round(c["TCC_MISS_sum"] * 128 / 1e9, 2),
        "fabric_TB_per_s": round(c["TCC_MISS_sum"] * 128 / 1e12 / (ms / 1e3), 2),
        "fabric_ceiling_TB_per_s": round(rate[table_mb] * 256 / 1e3, 2),
        "fetch_size_GB_x2": round(c["FETCH_SIZE"] * 1024 * 2 / 1e9, 2), "write_size_GB": round(c["WRITE_SIZE"] * 1024 / 1e9, 2),
        "additive_gather_model_ms": round(t_model, 3), "model_over_measured": round(t_model / ms, 3),
    }
    if floor_key is None:        # the E-step is bound by its 25.7 GB store stream, not by gathers: no gather model
        entry.pop("additive_gather_model_ms"); entry.pop("model_over_measured")
        entry["store_stream_TB_per_s"] = round(c["WRITE_SIZE"] * 1024 / 1e12 / (ms / 1e3), 2)
    if floor_key:
        entry["gather_only_same_traversal_ms"] = floor[floor_key]
        entry["gather_only_over_measured"] = round(floor[floor_key] / ms, 3)
    rows.append(entry)
out = {"ceilings": {"valu_v_fma_f32_wave_instr_per_simd_per_us": valu,
                    "random_256B_row_gathers_rows_per_ns_by_table_MB": rate,
                    "gather_only_floors_ms": floor},
       "kernels": rows}
json.dump(out, open(os.path.join(P, "r03_speed_of_light.json"), "w"), indent=1)
with open(os.path.join(P, "r03_speed_of_light.md"), "w") as f:
    f.write("# Speed of light, round 3 (config 3: 1 M docs x 100 k words, 100.4 M nnz, k = 64; 1x MI355X)\n\n")
    f.write("Sources: `r03_rocprofv3_kernel_stats_bench_cfg3.csv` (durations), `r03_rocprofv3_pmc_sq_tcc_cfg3.json` (counters, "
            "separate --pmc passes), `r03_sol_probe.jsonl` (ceilings measured by `tools/sol/sol_probe`).  Generated by "
            "`tools/sol/speed_of_light.py`.\n\n")
    f.write("Measured ceilings: VALU issue %.0f wave-instructions per SIMD per us (saturating `v_fma_f32`, 8 waves per SIMD; "
            "`v_pk_fma_f32` issues at 0.59x of that, i.e. 1.19x the flops: packed math is not a 2x lever on this part); random "
            "256-byte row gathers %.1f rows/ns out of a 2 MB table (L2 hits), %.1f / %.1f / %.1f rows/ns out of 25 MB / 256 MB / "
            "2 GB tables (L2 misses served by the Infinity Cache / HBM) = %.2f / %.2f / %.2f TB/s.\n\n"
            % (valu, rate[2], rate[25], rate[256], rate[2048], rate[25] * .256, rate[256] * .256, rate[2048] * .256))
    f.write("| kernel | rocprofv3 avg | VALU issue (of ceiling) | L2 requests (hit rate) | L2 request rate (of ceiling) | "
            "L2-miss traffic, rate (of ceiling) | additive gather model (of measured) | same traversal, arithmetic removed (of measured) |\n")
    f.write("|---|---|---|---|---|---|---|---|\n")
    for e in rows:
        f.write("| %s | %.3f ms | %.0f /SIMD/us (%.0f %%) | %.1f M (%.0f %%) | %.0f G/s (%.0f %%) | %.2f GB, %.2f TB/s (%.0f %%) | %s | %s |\n" % (
            e["kernel"], e["rocprofv3_avg_ms"], e["valu_issue_per_simd_per_us"], 100 * e["valu_issue_frac"],
            e["l2_requests_per_launch"] / 1e6, 100 * e["l2_hit_rate"], e["l2_requests_G_per_s"],
            100 * e["l2_requests_G_per_s"] / e["l2_request_ceiling_G_per_s"], e["l2_miss_bytes_GB"], e["fabric_TB_per_s"],
            100 * e["fabric_TB_per_s"] / e["fabric_ceiling_TB_per_s"],
            ("%.2f ms (%.0f %%)" % (e["additive_gather_model_ms"], 100 * e["model_over_measured"])) if "model_over_measured" in e else
            "-- (store-bound: %.2f TB/s of P written; non-temporal fill of the same buffer 6.0-6.7 TB/s)" % e["store_stream_TB_per_s"],
            ("%.2f ms (%.0f %%)" % (e["gather_only_same_traversal_ms"], 100 * e["gather_only_over_measured"])) if "gather_only_same_traversal_ms" in e else "--"))
    f.write("\nReading: a CU's gather pipeline serves L2 hits and L2 misses one after the other, so the time a pass needs for its "
            "gathers alone is `hit rows / hit rate + miss rows / miss rate`; both fused passes run AT that sum (document pass, "
            "98 %) or below it (column pass: hits overlap part of the misses), and the column pass within 5 % of the same traversal "
            "with every arithmetic instruction removed (the document pass within 14 % while the column tail runs beside it, within "
            "9 % alone: 1.485 against 1.35-1.38 ms in the probe).  Named bound: **the rate at which a CU's L1/L2 path delivers gathered 256-byte factor rows** "
            "(hits) plus **the rate at which L2 misses are served over the fabric** (misses).  VALU issue is the second resource "
            "in line (document pass ~80 % of the measured ceiling) -- removing thresholding or packing the arithmetic moves "
            "nothing while the gathers are the longer pole.\n")
print(open(os.path.join(P, "r03_speed_of_light.md")).read())
