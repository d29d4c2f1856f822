"""bench.py's static contract, checked without a GPU: the algorithmic-byte formulas reproduce the figures
SURVEY.md section 8d quotes, the config table is BASELINE.json's, the argument defaults are the ones the
driver relies on, and the module can be imported without touching the oracle or the device."""
import ast
import importlib.util
import json
import os

from conftest import ROOT


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)                      # defines functions only; main() is guarded
    return mod


def test_algorithmic_bytes_match_survey_section_8d():
    b = _bench()
    n, m, nnz, k = 1_000_000, 100_000, 100_000_000, 64         # config 3 at the nominal nnz
    gb = lambda kind: b.algorithmic_bytes(kind, n, m, nnz, k) / 1e9
    assert abs(gb("e_step") - 26.29) < 0.01                   # "B_E 26.29 GB (3.29 ms at 8 TB/s)"
    assert abs(gb("m_step_p") - 26.69) < 0.01                 # "B_M 26.69 GB"
    assert abs(gb("fused") - 1.37) < 0.01                     # "B_EM 1.37 GB"
    n, m, nnz, k = 100_000, 50_000, 10_000_000, 32            # config 2
    assert abs(b.algorithmic_bytes("e_step", n, m, nnz, k) / 1e9 - 1.34) < 0.005
    assert abs(b.algorithmic_bytes("fused", n, m, nnz, k) / 1e9 - 0.119) < 0.001
    n, m, nnz, k = 5_000_000, 200_000, 500_000_000, 128       # config 5
    assert abs(b.algorithmic_bytes("e_step", n, m, nnz, k) / 1e9 - 260.7) < 0.1
    assert b.HBM_PEAK_GBS == 8000.0
    assert set(b.KERNEL_KIND.values()) <= {"e_step", "m_step_p", "loglik", "fused", "fused_col"}


def test_config_table_is_baseline_json():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert b.CONFIGS[2]["n"] == 100_000 and b.CONFIGS[2]["m"] == 50_000 and b.CONFIGS[2]["nnz"] == 10_000_000 and b.CONFIGS[2]["k"] == 32
    assert b.CONFIGS[3]["n"] == 1_000_000 and b.CONFIGS[3]["m"] == 100_000 and b.CONFIGS[3]["nnz"] == 100_000_000 and b.CONFIGS[3]["k"] == 64
    assert b.CONFIGS[5]["n"] == 5_000_000 and b.CONFIGS[5]["m"] == 200_000 and b.CONFIGS[5]["nnz"] == 500_000_000 and b.CONFIGS[5]["k"] == 128
    assert b.CONFIGS[4]["n"] == b.CONFIGS[1]["n"] == 18_846 and b.CONFIGS[4]["k"] == 20 and b.CONFIGS[4]["ensemble_runs"] == 32
    assert "n_runs=32" in base["configs"][3] and "n_components=20" in base["configs"][3]
    assert "100k docs" in base["configs"][1] and "1M docs" in base["configs"][2] and "5M docs" in base["configs"][4]
    assert base["metric"].startswith("EM iterations/sec")


def test_defaults_and_oracle_usage():
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    defaults = {}
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and getattr(node.func, "attr", "") == "add_argument" and node.args:
            name = node.args[0].value if isinstance(node.args[0], ast.Constant) else None
            for kw in node.keywords:
                if kw.arg == "default" and isinstance(kw.value, ast.Constant):
                    defaults[name] = kw.value.value
    assert defaults["--gpus"] == 1 and defaults["--steps"] == 50 and defaults["--warmup"] == 5 and defaults["--config"] == 3
    # the oracle is only ever touched inside cpu_baseline()
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "cpu_baseline")
    inside = ast.get_source_segment(src, fn)
    assert "oracle" in inside
    outside = src.replace(inside, "")
    code_lines = [l for l in outside.splitlines() if "import" in l and "oracle" in l]
    assert not code_lines, code_lines


def test_the_printed_line_is_compact_and_carries_the_timed_loop_roofline():
    """The driver keeps a 2000-character tail of stdout and parses `roofline` out of the line: the default line is the compact one
    (compact_line) -- contract keys, `roofline` with the north-star kernel (k_e_step, labelled as the separate materialised leg)
    AND `timed_loop` = the dominant kernel of the loop `value` times, `cpu_baseline`, one short object per other BASELINE
    configuration and ensemble leg -- built here from a committed full record of a real run."""
    b = _bench()
    out = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_cfg3_1gpu_final_tree.json")))
    d = out["roofline_dominant_fused"]
    out["roofline"]["leg"] = "materialised (separate from value)"
    out["roofline"]["timed_loop"] = {"kernel": d["kernel"], "avg_ms": d["avg_launch_ms"], "algorithmic_GB": d["algorithmic_GB_per_launch"],
                                     "frac": d["frac"], "traffic_GB": 8.11, "traffic_ratio": 7.3, "frac_on_traffic": 0.62,
                                     "share_of_ms_per_step": 0.51}
    for leg in ("config1", "ensemble_20ng_shape"):
        out["other_configs"][leg]["corpus"] = b.corpus_kind(b.TOPICAL_20NG)
    line = b.compact_line(out, "gpurun_out/bench_full_cfg3_n1.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 2000, len(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    r = line["roofline"]
    assert r["kernel"] == "k_e_step" and r["leg"].startswith("materialised") and r["bound"] == "hbm" and 0 < r["frac"] < 1
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    t = r["timed_loop"]
    assert t["kernel"].startswith("k_") and 0 < t["frac"] < 1 and t["avg_ms"] > 0 and t["traffic_ratio"] > 1
    assert set(line["other_configs"]) >= {"config1", "config2", "config5", "config3_topical", "ensemble_20ng_shape"}
    assert line["other_configs"]["config1"]["corpus"] == "topical" and line["config"]["corpus"].startswith("independent")
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1
    # the 20-Newsgroups configurations default to the topical stand-in, config 3 to SURVEY 8d's generator
    import types
    assert b.corpus_options(types.SimpleNamespace(topics=-1, config=4))["topics"] == 20
    assert b.corpus_options(types.SimpleNamespace(topics=-1, config=3)) == {}
    assert b.corpus_options(types.SimpleNamespace(topics=0, config=1)) == {}
