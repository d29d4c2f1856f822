#!/usr/bin/env python3
"""Experiment: does re-ordering the DOCUMENTS make the column pass' gathers of
P(z|d) rows more local?  Documents are sorted by a min-hash over their mid-frequency words (document
frequency in [1e-4, 1e-1]) so that documents sharing such a word become neighbours; control: a random
permutation.  EM results are invariant under a document permutation (tests: test_document_permutation_
equivariance), so only the kernel times matter.
    python tools/experiments/doc_reorder.py [--config 3]"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from enstop_amd.engine import Engine, PLSA_FUSED  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--config", type=int, default=3)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
n, m, k = cfg["n"], cfg["m"], cfg["k"]
eng = Engine(0)
eng.generate_synthetic(n, m, cfg["nnz"], seed=0)
X = eng.download_active_csr()
U0, V0 = bench.init_factors(n, m, k, 42)

df = np.bincount(X.indices, minlength=m) / float(n)
mid = (df >= 1e-4) & (df <= 1e-1)
h = (X.indices.astype(np.uint64) * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
h[~mid[X.indices]] = np.uint64(0xFFFFFFFF)
key = np.minimum.reduceat(h, X.indptr[:-1].astype(np.int64))
orders = {"as generated": None, "min-hash over mid-frequency words": np.argsort(key, kind="stable"),
          "random permutation": np.random.RandomState(0).permutation(n)}
# a second key: the two smallest hashes (tighter clusters)
for name, order in orders.items():
    Y = X if order is None else X[order]
    eng.upload_csr(Y)
    eng.set_factors(U0 if order is None else U0[order], V0)
    eng.fit(None, n_iter=5, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    eng.timing(True); eng.timing_reset()
    it, ll = eng.fit(None, n_iter=a.steps, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    rep = {kk: round(v[1] / v[0], 4) for kk, v in eng.timing_report().items() if "pass" in kk}
    eng.timing(False)
    print(json.dumps({"config": a.config, "document_order": name, "avg_ms": rep, "ll_last": float(ll[-1])}), flush=True)
