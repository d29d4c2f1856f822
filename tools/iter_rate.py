#!/usr/bin/env python3
"""EM iterations/s of one BASELINE config WITHOUT per-kernel events (they cost ~13 % at 0.2 ms per iteration):
    python tools/iter_rate.py --config 1 [--steps 200] [--flags fused|materialised] [--events]
Knobs are read from the environment by the engine (PLSA_E_SEG, PLSA_COL_SEG, ...)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from enstop_amd.engine import Engine, PLSA_FUSED  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=1)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--flags", default="fused")
ap.add_argument("--events", action="store_true")
ap.add_argument("--estep", action="store_true", help="time the materialising E-step kernel instead of fits")
ap.add_argument("--tag", default="")
ap.add_argument("--shape", default="", help="n,m,nnz,k of a synthetic corpus instead of a BASELINE config")
ap.add_argument("--per-test", type=int, default=10, help="n_iter_per_test of the timed fit (a likelihood test = one host round trip)")
ap.add_argument("--no-zero-arm", action="store_true", help="stop test without the `change == 0` arm (timing experiments)")
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
if a.shape:
    n_, m_, nnz_, k_ = (int(float(x)) for x in a.shape.split(","))
    cfg = dict(n=n_, m=m_, nnz=nnz_, k=k_)
eng = Engine(0)
nnz = eng.generate_synthetic(cfg["n"], cfg["m"], cfg["nnz"], seed=0)
U0, V0 = bench.init_factors(cfg["n"], cfg["m"], cfg["k"], 42)
flags = (PLSA_FUSED if a.flags == "fused" else 0) | (16 if a.no_zero_arm else 0)
eng.set_factors(U0, V0)
if a.estep:
    eng.timing(True)
    eng.e_step(1e-32, want_host_copy=False)
    best = None
    for _ in range(a.reps):
        eng.timing_reset()
        for _ in range(10):
            eng.e_step(1e-32, want_host_copy=False)
        ms, cnt = eng.timing_get("k_e_step")
        best = ms / cnt if best is None else min(best, ms / cnt)
    b = bench.algorithmic_bytes("e_step", cfg["n"], cfg["m"], nnz, cfg["k"])
    print(json.dumps({"tag": a.tag, "config": a.config, "e_step_ms": round(best, 5), "GBps": round(b / 1e9 / (best / 1e3), 1),
                      "frac": round(b / 1e9 / (best / 1e3) / bench.HBM_PEAK_GBS, 4),
                      "env": {k: v for k, v in os.environ.items() if k.startswith("PLSA_")}}))
    sys.exit(0)
eng.fit(None, n_iter=10, n_iter_per_test=10, tolerance=0.0, flags=flags)
best = None
for _ in range(a.reps):
    eng.set_factors(U0, V0)
    eng.fit(None, n_iter=5, n_iter_per_test=10, tolerance=0.0, flags=flags)
    if a.events:
        eng.timing(True); eng.timing_reset()
    eng.synchronize()
    t0 = time.perf_counter()
    it, ll = eng.fit(None, n_iter=a.steps, n_iter_per_test=a.per_test, tolerance=0.0, flags=flags)
    eng.synchronize()
    dt = time.perf_counter() - t0
    assert it == a.steps
    best = dt if best is None else min(best, dt)
out = {"tag": a.tag, "config": a.config, "nnz": nnz, "steps": a.steps, "ms_per_iter": round(best / a.steps * 1e3, 5),
       "iter_per_s": round(a.steps / best, 1), "ll_last": float(ll[-1]),
       "env": {k: v for k, v in os.environ.items() if k.startswith("PLSA_")}}
if a.events:
    out["kernels"] = {k: round(v[1] / v[0], 5) for k, v in eng.timing_report().items()}
print(json.dumps(out))
