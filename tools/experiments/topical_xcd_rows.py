#!/usr/bin/env python3
"""Round 5: does a document order that follows the TOPICS pay once the document pass keeps each XCD on one contiguous
range of documents (PLSA_ROW_XCD=1)?  Topical corpus of config 3's shape; document orders: as generated / sorted by the
generator's ground-truth dominant topic (Engine.synthetic_dominant_topics) / random; each with the XCD-contiguous
document schedule off and on (the knob is read when a context is created: one process per setting).
    for x in 0 1; do PLSA_ROW_XCD=$x python tools/experiments/topical_xcd_rows.py; done"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from enstop_amd.engine import Engine, PLSA_FUSED  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--topics", type=int, default=64)
ap.add_argument("--alpha", type=float, default=0.1)
ap.add_argument("--background", type=float, default=0.25)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
n, m, k = cfg["n"], cfg["m"], cfg["k"]
eng = Engine(0)
nnz = eng.generate_synthetic(n, m, cfg["nnz"], seed=0, topics=a.topics, alpha=a.alpha, background=a.background)
lab = eng.synthetic_dominant_topics()
U0, V0 = bench.init_factors(n, m, k, 42)
orders = {"as generated": None, "sorted by ground-truth dominant topic": np.argsort(lab, kind="stable"),
          "random permutation": np.random.RandomState(0).permutation(n)}
for oname, order in orders.items():
    eng.bootstrap(None if order is None else order.astype(np.int64))
    eng.set_factors(U0 if order is None else U0[order], V0)
    eng.fit(None, n_iter=5, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    eng.synchronize()
    t0 = time.perf_counter()
    it, ll = eng.fit(None, n_iter=a.steps, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    eng.synchronize()
    dt = time.perf_counter() - t0
    eng.timing(True); eng.timing_reset()
    eng.fit(None, n_iter=10, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    rep = {kk: round(v[1] / v[0], 4) for kk, v in eng.timing_report().items() if "pass" in kk}
    eng.timing(False)
    print(json.dumps({"corpus": "t%d_a%g_b%g" % (a.topics, a.alpha, a.background), "PLSA_ROW_XCD": os.environ.get("PLSA_ROW_XCD", "0"),
                      "document_order": oname, "iter_per_s": round(it / dt, 1), "avg_ms": rep, "ll_last": float(ll[-1])}), flush=True)
