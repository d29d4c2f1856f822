#!/usr/bin/env python3
"""Round 5: every L2 / band / document-reordering conclusion of rounds 1-4 was drawn on a corpus
whose tokens are independent draws.  This script repeats the measurements on TOPICAL corpora
(plsa_generate_synthetic_topics) of config 3's shape:
  * EM iterations/s and per-pass times from random factors (what bench.py times) and after the factors have sharpened
  * document reordering: as generated / sorted by the dominant topic of P(z|d) after `--settle` iterations (device row
    gather: Engine.bootstrap(order)) / random permutation (control), all three from the SAME factors
    python tools/experiments/topical_corpus.py [--config 3] [--steps 30] [--settle 40]"""
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
ap.add_argument("--settle", type=int, default=40)
ap.add_argument("--corpora", default="independent,t64_a0.1_b0.25,t64_a0.05_b0.1,t256_a0.05_b0.1")
a = ap.parse_args()
cfg = bench.CONFIGS[a.config]
n, m, k = cfg["n"], cfg["m"], cfg["k"]
eng = Engine(0)


def timed(tag, extra):
    eng.fit(None, n_iter=3, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    eng.synchronize()
    t0 = time.perf_counter()
    it, _ = eng.fit(None, n_iter=a.steps, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    eng.synchronize()
    dt = time.perf_counter() - t0
    eng.timing(True); eng.timing_reset()
    eng.fit(None, n_iter=10, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    rep = {kk: round(v[1] / v[0], 4) for kk, v in eng.timing_report().items() if "pass" in kk}
    eng.timing(False)
    print(json.dumps(dict(extra, state=tag, iter_per_s=round(it / dt, 1), avg_ms=rep)), flush=True)


for name in a.corpora.split(","):
    if name == "independent":
        kw = {}
    else:
        t, al, bg = name.split("_")
        kw = dict(topics=int(t[1:]), alpha=float(al[1:]), background=float(bg[1:]))
    nnz = eng.generate_synthetic(n, m, cfg["nnz"], seed=0, **kw)
    U0, V0 = bench.init_factors(n, m, k, 42)
    extra = {"corpus": name, "nnz": nnz}
    eng.set_factors(U0, V0)
    timed("random factors, as generated", extra)
    eng.set_factors(U0, V0)
    eng.fit(None, n_iter=a.settle, n_iter_per_test=10, tolerance=0.0, flags=PLSA_FUSED)
    U, V = eng.get_factors()
    extra["zero_fraction_U"] = round(float((U == 0).mean()), 4)
    extra["mean_max_topic_share"] = round(float(U.max(axis=1).mean()), 4)
    timed("after %d iterations, as generated" % a.settle, extra)
    dom = U.argmax(axis=1)
    orders = {"sorted by dominant topic": np.argsort(dom, kind="stable"),
              "sorted by (dominant, second) topic": np.lexsort((np.argsort(-U, axis=1)[:, 1], dom)),
              "random permutation": np.random.RandomState(0).permutation(n)}
    for oname, order in orders.items():
        eng.bootstrap(order.astype(np.int64))
        eng.set_factors(U[order], V)
        timed("after %d iterations, %s" % (a.settle, oname), extra)
    eng.bootstrap(None)
